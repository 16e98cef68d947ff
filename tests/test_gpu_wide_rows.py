"""The wide-row binary32 frame-resident min-sum kernel (fused_float.hip, fused_ms_wide_kernel): frames whose
channel LLRs do not fit in LDS beside the sums and whose checks have 33 ... 48 slots -- configs[3]'s code,
32000.2240.3.105 (row weights 42 / 43), and random codes of the same shape.  Everything the decoder returns is
compared with the oracle's binary32 twin of dec() (test_code_min_sum_B.c:137-208): decoded bits, iteration
counts, syndromes, verdicts and the posterior sums, bit for bit."""
import numpy as np
import pytest

from conftest import code_path, make_frames

pytestmark = pytest.mark.gpu


def run_oracle(fn, oc, llrs, s, it):
    res = [fn(oc, llrs[f], s[f], it) for f in range(len(llrs))]
    return (np.stack([r["bits"] for r in res]).astype(np.uint8), np.array([r["iters"] for r in res], np.int32),
            np.stack([r["synd"] for r in res]).astype(np.uint8), np.array([r["rc"] for r in res], np.int32),
            np.stack([r["post"] for r in res]))


def check(out, ref, tag, llr=True):
    assert np.array_equal(out["status"], ref[3]), f"{tag}: status differs"
    assert np.array_equal(out["iters"], ref[1]), f"{tag}: iteration counts differ"
    assert np.array_equal(out["bits"], ref[0]), f"{tag}: decoded bits differ"
    assert np.array_equal(out["synd"], ref[2]), f"{tag}: syndromes differ"
    nc = ref[3] != 0
    if llr and nc.any():
        assert np.array_equal(out["llr"][nc], ref[4][nc].astype(np.float32)), f"{tag}: posterior sums differ"


def mixed_frames(oracle, oc, seed):
    """Frames that stop after different iteration counts, frames that never stop, equal and distinct magnitudes,
    magnitudes that reach the cap of 200, zero and non-zero syndromes."""
    parts = [make_frames(oracle, oc, 6, 0.0005, seed, "bsc"), make_frames(oracle, oc, 5, 0.0009, seed + 1, "bsc"),
             make_frames(oracle, oc, 6, None, seed + 2, "awgn", sigma=0.38), make_frames(oracle, oc, 4, None, seed + 3, "awgn", sigma=0.43)]
    x = np.concatenate([p[0] for p in parts])
    s = np.concatenate([p[1] for p in parts])
    llr = np.concatenate([p[2] for p in parts])
    llr[-2:] *= 40.0        # |v2c| beyond the cap: min1 = min2 = 200
    llr[6, ::7] = 0.0       # exact zeros: sums of +-0.0, ties in the minimum
    return x, s.astype(np.uint8), llr.astype(np.float32)


@pytest.mark.parametrize("waves", [None, "12"])
def test_wide_kernel_on_the_32000_code(hip, oracle, monkeypatch, waves):
    name = "32000.2240.3.105.txt"
    if waves:
        monkeypatch.setenv("LDPC_HIP_FUSED_WAVES", waves)
    oc = oracle.load(code_path(name))
    x, s, llr32 = mixed_frames(oracle, oc, 41)
    it = 12
    ref = run_oracle(oracle.dec_minsum_f32, oc, llr32, s, it)
    assert 0 < (ref[3] != 0).sum() < len(llr32) and len(set(ref[1].tolist())) >= 4, "the case must mix verdicts and iteration counts"
    code = hip.Code.from_alist(code_path(name))
    want = ("bits", "iters", "synd", "status", "llr")
    for early in (True, False):
        dec = hip.Decoder(code, hip.ALG.MINSUM_F32, max_frames=64, early_exit=early, want_llr=True)
        dec.profile(True)
        check(dec.decode(llr32, s, it, want=want), ref, f"wide {waves} early_exit={early} f32")
        ms, launches = dec.profile_read()
        assert launches[2] >= 1 and launches[0] == 0 and launches[1] == 0, "the frame-resident kernel must have run, not the two-pass engine"
        dec.profile(False)
        check(dec.decode(llr32.astype(np.float64), s, it, want=want), ref, f"wide {waves} early_exit={early} f64 input")
        # a second call on the same decoder, fewer frames, no posterior asked for
        out = dec.decode(llr32[3:11], s[3:11], it, want=("bits", "iters", "synd", "status"))
        check(out, tuple(r[3:11] for r in ref), f"wide {waves} second call", llr=False)
        dec.close()
    # zero and one iteration
    for k in (0, 1):
        refk = run_oracle(oracle.dec_minsum_f32, oc, llr32[:8], s[:8], k)
        dec = hip.Decoder(code, hip.ALG.MINSUM_F32, max_frames=64, want_llr=True)
        check(dec.decode(llr32[:8], s[:8], k, want=want), refk, f"wide {waves} max_iter={k}")
        dec.close()
    # the two-pass engine on the same frames (cn_variant 2 selects it): same answers from a different program
    dec = hip.Decoder(code, hip.ALG.MINSUM_F32, max_frames=64, want_llr=True, cn_variant=2)
    check(dec.decode(llr32, s, it, want=want), ref, "two-pass on the wide case")


def random_wide_code(rng, n, m, dv=3):
    """A (dv, ~dv n / m)-regular random code: checks filled evenly, no double edges."""
    H = np.zeros((m, n), np.uint8)
    load = np.zeros(m, np.int64)
    for v in range(n):
        # the dv lightest checks among a random sample keep the row weights within one or two of each other
        cand = rng.choice(m, size=24, replace=False)
        pick = cand[np.argsort(load[cand], kind="stable")[:dv]]
        H[pick, v] = 1
        load[pick] += 1
    return H


@pytest.mark.parametrize("n,m,seed", [(31488, 2200, 5), (32064, 2230, 6)])
def test_wide_kernel_on_random_codes(hip, oracle, tmp_path, n, m, seed):
    """Other codes that land on the kernel's build (3 checks and 32 variables per thread, 43 / 44 slots):
    different row-weight mixes, pad slots in every check or in none, a last round with few checks."""
    rng = np.random.default_rng(seed)
    H = random_wide_code(rng, n, m)
    rw = H.sum(axis=1)
    assert 33 <= rw.max() <= 44 and rw.max() >= 43, f"row weights {rw.min()}..{rw.max()} miss the kernel's shape"
    path = str(tmp_path / "wide.txt")
    hip.alist.matrix_to_alist(H, path)
    oc = oracle.load(path)
    x, s, llr = make_frames(oracle, oc, 10, 0.004, seed, "awgn", sigma=0.41)
    llr32 = llr.astype(np.float32)
    s = s.astype(np.uint8)
    it = 10
    ref = run_oracle(oracle.dec_minsum_f32, oc, llr32, s, it)
    code = hip.Code.from_alist(path)
    dec = hip.Decoder(code, hip.ALG.MINSUM_F32, max_frames=64, want_llr=True)
    dec.profile(True)
    check(dec.decode(llr32, s, it, want=("bits", "iters", "synd", "status", "llr")), ref, f"random wide code {n}x{m}")
    ms, launches = dec.profile_read()
    assert launches[2] >= 1 and launches[0] == 0, "the frame-resident kernel must have run"
