"""GPU parity of the staircase (IRA / DVB-S2) frame-resident Q8.8 decoder, chain_q88.hip,
against the CPU oracle of test_code_min_sum_B_fixed.c:192-316 -- bit-exact on decoded bits,
iteration counts, syndromes, verdicts and posteriors, through the C-ABI.

The kernel serves codes with a weight-2 variable between consecutive checks: DVB-S2 0.5 by
default (its 64 800 variables do not fit the LDS sums of the other frame-resident decoder), and
small random codes of the same build here, where LDPC_HIP_CHAIN=2 prefers it.
"""
import numpy as np
import pytest

from conftest import code_path, make_frames, to_q88
from test_gpu_binary import check_exact, run_oracle

pytestmark = pytest.mark.gpu


def staircase_code(rng, m, k, info_w=(5, 5), holes=0, tail=True, six=0):
    """H = [A | T]: A has info_w[0]..info_w[1] ones per row in k information columns, T is the
    dual diagonal (column j: rows j and j + 1), minus `holes` columns that get a third one
    elsewhere (no longer weight 2: they live in LDS), last column weight 1 if `tail`;
    `six` rows get a sixth information bit."""
    H = np.zeros((m, k + m), np.uint8)
    for j in range(m):
        w = int(rng.integers(info_w[0], info_w[1] + 1))
        H[j, rng.choice(k, size=w, replace=False)] = 1
    for j in rng.choice(m, size=six, replace=False):
        free = np.flatnonzero(H[j, :k] == 0)
        need = 6 - int(H[j, :k].sum())
        H[j, rng.choice(free, size=need, replace=False)] = 1
    for v in np.flatnonzero(H[:, :k].sum(axis=0) == 0):      # every information bit is checked
        H[int(rng.integers(0, m)), v] = 1
    for j in range(m):
        H[j, k + j] = 1
        if j + 1 < m:
            H[j + 1, k + j] = 1
    if not tail:
        H[0, k + m - 1] = 1
    for j in rng.choice(m - 3, size=holes, replace=False):
        H[j + 2, k + j] = 1      # a third one, two rows down
    return H


def fused_ran(dec, fn):
    dec.profile(True)
    out = fn()
    _, cnt = dec.profile_read()
    dec.profile(False)
    assert cnt[2] > 0 and cnt[0] == 0 and cnt[1] == 0, "the frame-resident kernel did not run"
    return out


SHAPES = [
    # m, k, info weights, holes, tail, rows with six, frames, p, iterations
    (250, 200, (5, 5), 0, True, 0, 40, 0.05, 12),
    (256, 300, (2, 5), 7, True, 5, 40, 0.04, 15),
    (1500, 1200, (3, 5), 20, False, 11, 16, 0.05, 20),
    (4096, 3000, (4, 5), 0, True, 3, 6, 0.06, 25),
]


@pytest.mark.parametrize("m,k,iw,holes,tail,six,F,p,it", SHAPES)
@pytest.mark.parametrize("kind", ["bsc", "awgn"])
def test_staircase_q88_random_codes(hip, oracle, tmp_path, monkeypatch, m, k, iw, holes, tail, six, F, p, it, kind):
    rng = np.random.default_rng(m * 7 + k)
    H = staircase_code(rng, m, k, iw, holes, tail, six)
    path = str(tmp_path / "stair.alist")
    hip.alist.matrix_to_alist(H, path)
    oc = oracle.load(path)
    x, s, llr = make_frames(oracle, oc, F, p, 5, kind, sigma=0.8)
    q, raw = to_q88(llr)
    ref = run_oracle(lambda c, l, sy, n_it: oracle.dec_minsum_fixed(c, l, sy, n_it, None, oc.m + 1), oc, q, s, it)
    monkeypatch.setenv("LDPC_HIP_CHAIN", "2")
    code = hip.Code.from_alist(path)
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=64, early_exit=True, want_llr=True)
    su = s.astype(np.uint8)
    want = ("bits", "iters", "synd", "status", "llr")
    for inp in (raw, q, llr.astype(np.float32).astype(np.float64)):
        ref_i = ref
        if inp.dtype == np.float64 and inp is not q:      # off-grid doubles: the oracle sees their float_to_fix
            qq, _ = to_q88(inp)
            ref_i = run_oracle(lambda c, l, sy, n_it: oracle.dec_minsum_fixed(c, l, sy, n_it, None, oc.m + 1), oc, qq, s, it)
        out = fused_ran(dec, lambda: dec.decode(inp, su, it, want=want))
        check_exact(out, ref_i, f"staircase {m}x{k + m} {kind} {inp.dtype}")
        nc = ref_i[3] != 0
        if nc.any():
            got, exp = out["llr"][nc].astype(np.float64), ref_i[4][nc]
            bad = np.argwhere(got != exp)
            assert len(bad) == 0, f"posteriors differ at (frame, variable) {bad[:8].tolist()}: {got[got != exp][:8]} vs {exp[got != exp][:8]}"
    # the two-pass engine decodes the same frames the same way
    monkeypatch.setenv("LDPC_HIP_CHAIN", "0")
    two = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=64, early_exit=True, cn_variant=2)
    check_exact(two.decode(raw, su, it), ref, "two-pass")


@pytest.mark.parametrize("kind,p,sigma,it,F", [("bsc", 0.06, None, 12, 5), ("awgn", 0.0, 0.87, 40, 6), ("awgn", 0.0, 1.1, 8, 3)])
def test_staircase_q88_dvb_s2(hip, oracle, tmp_path, kind, p, sigma, it, F):
    """configs[2]'s code, Q8.8 min-sum: frames that converge (after different iteration counts)
    and frames that do not; the default decoder for this code is the staircase kernel."""
    path = str(tmp_path / "padded.alist")
    hip.alist.normalise_alist(code_path("dvb_H_0.5.txt"), path, padded=True)
    oc = oracle.load(path)
    x, s, llr = make_frames(oracle, oc, F, p, 77, kind, sigma=sigma)
    q, raw = to_q88(llr)
    ref = run_oracle(lambda c, l, sy, n_it: oracle.dec_minsum_fixed(c, l, sy, n_it, None, oc.m + 1), oc, q, s, it)
    print(f"dvb q88 {kind}: oracle (rc, iters) {list(zip(ref[3], ref[1]))}")
    code = hip.Code.from_alist(code_path("dvb_H_0.5.txt"))
    su = s.astype(np.uint8)
    want = ("bits", "iters", "synd", "status", "llr")
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=16, early_exit=True, want_llr=True)
    for inp in (raw, q):
        out = fused_ran(dec, lambda: dec.decode(inp, su, it, want=want))
        check_exact(out, ref, f"dvb staircase {kind} {inp.dtype}")
        nc = ref[3] != 0
        if nc.any():
            assert np.array_equal(out["llr"][nc].astype(np.float64), ref[4][nc])
    dec2 = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=16, early_exit=False)
    check_exact(dec2.decode(raw, su, it), ref, f"dvb staircase no-exit {kind}")
    # more frames than workgroups in flight, against the two-pass engine
    rng = np.random.default_rng(3)
    many = np.tile(raw, (60, 1))[:300].copy()
    flip = rng.random(many.shape) < 0.002
    many[flip] = -many[flip]
    sm = np.tile(su, (60, 1))[:300]
    big = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=512, early_exit=True)
    two = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=512, early_exit=True, cn_variant=2)
    a = fused_ran(big, lambda: big.decode(many, sm, it))
    b = two.decode(many, sm, it)
    for key in ("bits", "iters", "synd", "status"):
        assert np.array_equal(a[key], b[key]), key
