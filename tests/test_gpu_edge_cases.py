"""Edge cases of the batch boundary (ldpc_decode_batch through the C ABI): no frames, one
frame, more frames than the decoder's pass size with a ragged tail, max_iter = 0 (dec() with
loop_max 0: only the hard decision of the input and its syndrome; the fixed-point variant's
"already a codeword" pre-check, test_code_min_sum_B_fixed.c:198-207), bad arguments refused
by code with the decoder still usable afterwards."""
import ctypes as C

import numpy as np
import pytest

from conftest import code_path, make_frames, to_q88
from test_gpu_binary import check_exact, run_oracle
from test_zz_no_alloc import _run as _run_on_device

pytestmark = pytest.mark.gpu
NAME = "204.33.484.txt"


def _ref_q88(oracle, oc, q, s, it):
    return [oracle.dec_minsum_fixed(oc, q[f], s[f], it, None, oc.m + 1) for f in range(len(q))]


def _same(out, ref, tag):
    assert np.array_equal(out["status"], [r["rc"] for r in ref]), tag
    assert np.array_equal(out["iters"], [r["iters"] for r in ref]), tag
    assert np.array_equal(out["bits"], np.stack([r["bits"] for r in ref])), tag
    assert np.array_equal(out["synd"], np.stack([r["synd"] for r in ref])), tag


def test_no_frames_one_frame_and_ragged_passes(hip, oracle):
    oc = oracle.load(code_path(NAME))
    code = hip.Code.from_alist(code_path(NAME))
    F = 64 * 3 + 5
    x, s, llr = make_frames(oracle, oc, F, 0.05, 77, "bsc")
    q, raw = to_q88(llr)
    ref = _ref_q88(oracle, oc, q, s, 20)
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=64)          # 4 passes, the last one of 5 frames
    assert dec.max_frames >= 64
    out = dec.decode(raw[:0], s[:0].astype(np.uint8), 20)
    assert out["bits"].shape == (0, code.n) and out["iters"].shape == (0,)
    _same(dec.decode(raw[:1], s[:1].astype(np.uint8), 20), ref[:1], "one frame")
    _same(dec.decode(raw, s.astype(np.uint8), 20), ref, "ragged passes")
    # outputs are optional, one by one
    only = dec.decode(raw, s.astype(np.uint8), 20, want=("iters",))
    assert list(only) == ["iters"] and np.array_equal(only["iters"], [r["iters"] for r in ref])
    # zero syndrome by NULL == explicit zeros
    z = np.zeros((F, code.m), np.uint8)
    a, b = dec.decode(raw, None, 20), dec.decode(raw, z, 20)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("alg_name", ["MINSUM_Q88", "MINSUM_F64", "MINSUM_F32", "SPA_F64", "SPA_F32", "SPA_TABLE_Q14"])
def test_zero_iterations(hip, oracle, alg_name):
    """loop_max = 0: no message passing; bits = (q0 < 0), synd = H bits; verdict 0 only where the
    fixed-point pre-check applies (the other dec() variants fall through to `return 1`)."""
    oc = oracle.load(code_path(NAME))
    code = hip.Code.from_alist(code_path(NAME))
    x, s, llr = make_frames(oracle, oc, 70, 0.004, 5, "bsc")           # many frames arrive error-free
    q, raw = to_q88(llr)
    dec = hip.Decoder(code, getattr(hip.ALG, alg_name), max_frames=128)
    inp = {"MINSUM_Q88": raw, "MINSUM_F32": llr.astype(np.float32), "SPA_F32": llr.astype(np.float32),
           "SPA_TABLE_Q14": oracle.float2fix14(llr)}.get(alg_name, llr)
    out = dec.decode(inp, s.astype(np.uint8), 0)
    hard = (llr < 0).astype(np.uint8)
    synd = np.stack([oracle.enc(oc, hard[f].astype(np.int32)) for f in range(len(hard))]).astype(np.uint8)
    assert np.array_equal(out["bits"], hard) and np.array_equal(out["synd"], synd)
    assert (out["iters"] == 0).all()
    clean = (synd == s).all(axis=1)
    assert clean.any() and (~clean).any()
    if alg_name == "MINSUM_Q88":
        _same(out, _ref_q88(oracle, oc, q, s, 0), "q88 loop_max 0")
        assert np.array_equal(out["status"] == 0, clean)
    else:
        assert (out["status"] != 0).all()


def test_bad_arguments_are_refused_and_the_decoder_survives(hip, oracle):
    L = hip.load_library()
    oc = oracle.load(code_path(NAME))
    code = hip.Code.from_alist(code_path(NAME))
    x, s, llr = make_frames(oracle, oc, 8, 0.05, 3, "bsc")
    q, raw = to_q88(llr)
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=64)
    good = dec.decode(raw, s.astype(np.uint8), 10)
    vp = C.c_void_p
    p = raw.ctypes.data_as(vp)
    bits = np.zeros((8, code.n), np.uint8)
    call = lambda h, nf, ptr, dt, it: L.ldpc_decode_batch(h, nf, ptr, dt, None, it, bits.ctypes.data_as(vp), None, None, None, None)
    assert call(dec._h, 8, None, hip.DT.I16, 10) < 0          # no LLRs
    assert call(dec._h, 8, p, 99, 10) < 0                     # unknown element type
    assert call(dec._h, -1, p, hip.DT.I16, 10) < 0            # negative frame count
    assert call(dec._h, 8, p, hip.DT.I16, -1) < 0             # negative loop_max
    assert call(None, 8, p, hip.DT.I16, 10) < 0               # no decoder
    assert L.ldpc_last_error()
    with pytest.raises(ValueError):
        dec.decode(raw.reshape(-1)[:-1], None, 10)            # not a whole number of frames
    again = dec.decode(raw, s.astype(np.uint8), 10)
    for k in good:
        assert np.array_equal(good[k], again[k]), k
    with pytest.raises(hip.LdpcError):
        hip.Code.from_alist(code_path("no_such_file.txt"))    # exit(-2) in the reference (:216-219)
    with pytest.raises(hip.LdpcError):
        hip.Decoder(code, 99)                                 # unknown algorithm


def test_gf_no_frames_and_zero_iterations(hip, oracle):
    """Same edges on the GF(4) Walsh-Hadamard decoder, inputs = the committed vectors of the
    compiled reference program (tests/golden/ref_ntt_q4.npz)."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "ref_ntt_q4.npz"))
    fname = "q4.sp.9000.6000.4500.1"
    code = hip.Code.from_alist(code_path(fname))
    oc = oracle.load(code_path(fname))
    dec = hip.Decoder(code, hip.ALG.GF2P_WHT, max_frames=2)            # 4 frames: two passes
    lf, s = g["logfna"].astype(np.int32), g["s"]
    out = dec.decode(lf[:0], s[:0], 5)
    assert out["bits"].shape == (0, code.n) and out["status"].shape == (0,)
    out = dec.decode(lf, s, 0)
    ref = [oracle.dec_gf2p_wht(oc, 2, lf[f], s[f].astype(np.int32), 0) for f in range(len(lf))]
    assert np.array_equal(out["bits"], np.stack([r["x"] for r in ref]))
    assert np.array_equal(out["synd"], np.stack([r["z"] for r in ref]))
    assert np.array_equal(out["status"] != 0, np.array([r["rc"] != 0 for r in ref]))
    assert np.array_equal(out["iters"], np.array([r["iters"] for r in ref]))
    full = dec.decode(lf, s, int(g["maxit"]))
    assert np.array_equal(full["bits"], g["xhat"]) and np.array_equal(full["status"] != 0, g["rc"] != 0)


@pytest.mark.parametrize("alg", ["MINSUM_Q88", "SPA_F32", "MINSUM_F64"])
def test_two_stage_decode_equals_one_stage(hip, oracle, monkeypatch, alg):
    """Two-pass engine, early exit, a batch large enough for the two-stage schedule (BinaryEngine::decode):
    the first stage stops once at most 1/16 of the frames are open, those are decoded again as a dense
    batch.  Every output must equal the one-stage decode's, and (Q8.8) the oracle's."""
    name, F, it = "204.33.484.txt", 4500, 40
    oc = oracle.load(code_path(name))
    x, s, llr = make_frames(oracle, oc, F, 0.0, 23, "awgn", sigma=0.72)
    q, raw = to_q88(llr)
    su = s.astype(np.uint8)
    monkeypatch.setenv("LDPC_HIP_FUSED", "0")            # the two-pass engine for every algorithm
    code = hip.Code.from_alist(code_path(name))
    inp = raw if alg == "MINSUM_Q88" else (llr.astype(np.float32) if alg == "SPA_F32" else llr)
    outs, launches = {}, {}
    for mode in ("0", "1"):
        monkeypatch.setenv("LDPC_HIP_TWO_STAGE", mode)
        dec = hip.Decoder(code, getattr(hip.ALG, alg), max_frames=F, early_exit=True, want_llr=True)
        dec.profile(True)
        outs[mode] = _run_on_device(hip, dec, code, inp, su, it, False, want_llr=True)   # (the host-buffer entry decodes in chunks)
        _, cnt = dec.profile_read()
        launches[mode] = int(cnt[1])
        dec.close()
    one, two = outs["0"], outs["1"]
    open_frames = int((one["status"] != 0).sum())
    slow = int((one["iters"] > 12).sum())
    print(f"{alg}: {open_frames} frames fail, {slow} need more than 12 iterations; check passes launched: one stage {launches['0']}, two stages {launches['1']}")
    assert 0 < open_frames <= F // 16, "pick a noise level at which a few frames fail"
    # a failing frame keeps one stage going to the last iteration; two stages: a short first stage, then all of them again
    assert launches["0"] == it and it < launches["1"] < 2 * it - 7
    for key in ("status", "iters", "bits", "synd"):
        assert np.array_equal(one[key], two[key]), key
    assert np.array_equal(one["llr"], two["llr"], equal_nan=True)
    if alg == "MINSUM_Q88":
        pick = np.unique(np.concatenate([np.flatnonzero(one["iters"] > 6)[:80], np.arange(60)]))
        ref = run_oracle(lambda c, l, sy, k: oracle.dec_minsum_fixed(c, l, sy, k, None, oc.m + 1), oc, q[pick], s[pick], it)
        sub = {k: two[k][pick] for k in ("status", "iters", "bits", "synd")}
        check_exact(sub, ref, "two-stage q88")
