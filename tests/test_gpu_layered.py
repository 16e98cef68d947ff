"""Layered schedule of the Q8.8 min-sum decoder (ldpc_opts.layer_checks; SURVEY 8 f4, not in the
reference) against its CPU restatement ora_dec_minsum_fixed_layered(), bit for bit: decoded
bits, iteration counts, syndromes, status and posteriors.  The restatement itself is pinned to
the reference's flooding decoder in tests/test_oracle_golden.py (layer >= m)."""
import numpy as np
import pytest

from conftest import code_path, make_frames, to_q88
from test_gpu_binary import check_exact, run_oracle

pytestmark = pytest.mark.gpu

CASES = [
    # code, layer_checks, F, p (bsc), sigma (awgn), max_iter
    ("204.33.484.txt", 64, 200, 0.04, 0.75, 20),
    ("1057.244.3.txt", 64, 200, 0.014, 0.56, 15),
    ("1057.244.3.txt", 128, 100, 0.014, 0.56, 15),
    ("4095.737.3.txt", 128, 33, 0.009, 0.5, 12),
    ("4095.737.3.txt", 768, 20, 0.009, 0.5, 12),       # one layer: the reference's flooding schedule
    ("peg_16000_3_0.9.txt", 256, 10, 0.0016, 0.43, 10),
    ("peg_16000_3_0.9.txt", 1024, 10, 0.0016, 0.43, 10),
]


@pytest.mark.parametrize("name,layer,F,p,sigma,it", CASES)
@pytest.mark.parametrize("kind", ["bsc", "awgn"])
def test_layered_q88_equals_its_restatement(hip, oracle, name, layer, F, p, sigma, it, kind):
    oc = oracle.load(code_path(name))
    x, s, llr = make_frames(oracle, oc, F, p, 21, kind, sigma=sigma)
    q, raw = to_q88(llr)
    ref = run_oracle(lambda c, l, sy, k: oracle.dec_minsum_fixed_layered(c, l, sy, k, layer), oc, q, s, it)
    code = hip.Code.from_alist(code_path(name))
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=256, early_exit=True, want_llr=True, layer_checks=layer)
    for inp in (raw, llr.astype(np.float64)):
        out = dec.decode(inp, s.astype(np.uint8), it, want=("bits", "iters", "synd", "status", "llr"))
        check_exact(out, ref, f"layered {name} L={layer} {kind} {inp.dtype}")
    nc = ref[3] != 0
    if nc.any():
        assert np.array_equal(out["llr"][nc].astype(np.float64), ref[4][nc])
    if layer < oc.m or kind == "awgn":   # (un-normalised flooding min-sum ties on equal-magnitude BSC inputs)
        assert (ref[3] == 0).any(), "no frame converged: the case tests nothing"
    if layer >= oc.m:
        flood = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=256, want_llr=True)
        check_exact(flood.decode(raw, s.astype(np.uint8), it), ref, f"one layer = flooding, {name}")


def test_layered_needs_fewer_iterations_than_flooding(hip, oracle):
    """The point of the schedule: on the same frames the layered decoder converges in fewer
    iterations and fails no more often (1057.244.3, BSC)."""
    name = "1057.244.3.txt"
    oc = oracle.load(code_path(name))
    x, s, llr = make_frames(oracle, oc, 2048, 0.01, 22, "bsc")
    q, raw = to_q88(llr)
    code = hip.Code.from_alist(code_path(name))
    flood = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=2048).decode(raw, s.astype(np.uint8), 50)
    lay = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=2048, layer_checks=64).decode(raw, s.astype(np.uint8), 50)
    assert lay["status"].sum() <= flood["status"].sum()
    assert lay["iters"].mean() < 0.7 * flood["iters"].mean()
    ok = lay["status"] == 0
    assert np.array_equal(lay["synd"][ok], s[ok].astype(np.uint8))


def test_layered_option_is_refused_where_it_does_not_apply(hip):
    code = hip.Code.from_alist(code_path("1057.244.3.txt"))
    for kw in (dict(alg=hip.ALG.MINSUM_F32, layer_checks=64), dict(alg=hip.ALG.MINSUM_Q88, layer_checks=100),
               dict(alg=hip.ALG.MINSUM_Q88, layer_checks=2048)):
        with pytest.raises(hip.LdpcError):
            hip.Decoder(code, kw["alg"], layer_checks=kw["layer_checks"])
    big = hip.Code.from_alist(code_path("dvb_H_0.5.txt"))
    with pytest.raises(hip.LdpcError):
        hip.Decoder(big, hip.ALG.MINSUM_Q88, layer_checks=1024)      # 64 800 variables do not fit LDS
