"""GPU parity of the staircase binary32 min-sum decoder (stair_f32.hip) against the oracle's binary32 twin of
dec() (test_code_min_sum_B.c:137-208): decoded bits, iteration counts, syndromes, verdicts and posterior sums,
bit for bit, through the C-ABI.  DVB-S2 0.5 takes the kernel by default (its frame fits neither LDS layout of
fused_float.hip); small random accumulator-chain codes reach it under LDPC_HIP_CHAIN=2, with 256-thread
workgroups (LDPC_HIP_STAIR_BLOCK) so that they too have many rounds and same-round pairs."""
import numpy as np
import pytest

from conftest import code_path, make_frames
from test_gpu_chain import staircase_code, fused_ran
from test_gpu_wide_rows import run_oracle, check

pytestmark = pytest.mark.gpu

WANT = ("bits", "iters", "synd", "status", "llr")


def dvb_frames(oracle, oc, seed):
    """Frames that stop after different iteration counts and frames that do not; equal and distinct magnitudes,
    exact zeros, magnitudes beyond the cap; zero and non-zero syndromes."""
    parts = [make_frames(oracle, oc, 5, None, seed, "awgn", sigma=0.70), make_frames(oracle, oc, 4, None, seed + 1, "awgn", sigma=0.78),
             make_frames(oracle, oc, 3, 0.04, seed + 2, "bsc"), make_frames(oracle, oc, 2, 0.07, seed + 3, "bsc")]
    s = np.concatenate([p[1] for p in parts]).astype(np.uint8)
    llr = np.concatenate([p[2] for p in parts])
    llr[3, ::11] = 0.0
    llr[8] *= 60.0
    return s, llr.astype(np.float32)


def test_staircase_f32_on_dvb_s2(hip, oracle, tmp_path, monkeypatch):
    name = "dvb_H_0.5.txt"
    padded = str(tmp_path / "dvb_padded.txt")      # (the reference-style loader of the oracle needs the padded dialect)
    hip.alist.normalise_alist(code_path(name), padded, padded=True)
    oc = oracle.load(padded)
    s, llr32 = dvb_frames(oracle, oc, 77)
    it = 15
    ref = run_oracle(oracle.dec_minsum_f32, oc, llr32, s, it)
    print("verdicts", ref[3], "iterations", ref[1])
    assert 0 < (ref[3] != 0).sum() < len(llr32) and len(set(ref[1].tolist())) >= 4, "the case must mix verdicts and iteration counts"
    code = hip.Code.from_alist(code_path(name))
    for early in (True, False):
        dec = hip.Decoder(code, hip.ALG.MINSUM_F32, max_frames=64, early_exit=early, want_llr=True)
        check(fused_ran(dec, lambda: dec.decode(llr32, s, it, want=WANT)), ref, f"dvb early_exit={early} f32")
        check(dec.decode(llr32.astype(np.float64), s, it, want=WANT), ref, f"dvb early_exit={early} f64 input")
        out = dec.decode(llr32[2:9], s[2:9], it, want=("bits", "iters", "synd", "status"))
        check(out, tuple(r[2:9] for r in ref), "dvb second call", llr=False)
        dec.close()
    for k in (0, 1, 2):
        refk = run_oracle(oracle.dec_minsum_f32, oc, llr32[:6], s[:6], k)
        dec = hip.Decoder(code, hip.ALG.MINSUM_F32, max_frames=64, want_llr=True)
        check(dec.decode(llr32[:6], s[:6], k, want=WANT), refk, f"dvb max_iter={k}")
        dec.close()
    # the two-pass engine on the same frames: same answers from a different program
    monkeypatch.setenv("LDPC_HIP_CHAIN", "0")
    dec = hip.Decoder(code, hip.ALG.MINSUM_F32, max_frames=64, want_llr=True)
    check(dec.decode(llr32, s, it, want=WANT), ref, "two-pass on the dvb case")


SHAPES = [
    # m, k, info weights, holes, tail, rows with six, frames, sigma, iterations, threads
    (250, 200, (5, 5), 0, True, 0, 30, 0.75, 12, 256),
    (900, 300, (2, 5), 7, True, 5, 30, 0.8, 15, 256),
    (1500, 1200, (3, 5), 20, False, 11, 16, 0.7, 20, 256),
    (4096, 3000, (4, 5), 0, True, 3, 8, 0.7, 25, 256),
    (4096, 3000, (4, 5), 9, True, 3, 8, 0.7, 25, 1024),
]


@pytest.mark.parametrize("m,k,iw,holes,tail,six,F,sigma,it,threads", SHAPES)
@pytest.mark.parametrize("kind", ["bsc", "awgn"])
def test_staircase_f32_random_codes(hip, oracle, tmp_path, monkeypatch, m, k, iw, holes, tail, six, F, sigma, it, threads, kind):
    rng = np.random.default_rng(m * 11 + k + threads)
    H = staircase_code(rng, m, k, iw, holes, tail, six)
    path = str(tmp_path / "stair.alist")
    hip.alist.matrix_to_alist(H, path)
    oc = oracle.load(path)
    x, s, llr = make_frames(oracle, oc, F, 0.05, 5, kind, sigma=sigma)
    llr32 = llr.astype(np.float32)
    llr32[0, ::5] = 0.0
    s = s.astype(np.uint8)
    ref = run_oracle(oracle.dec_minsum_f32, oc, llr32, s, it)
    monkeypatch.setenv("LDPC_HIP_CHAIN", "2")
    monkeypatch.setenv("LDPC_HIP_STAIR_BLOCK", str(threads))
    code = hip.Code.from_alist(path)
    dec = hip.Decoder(code, hip.ALG.MINSUM_F32, max_frames=64, early_exit=True, want_llr=True)
    check(fused_ran(dec, lambda: dec.decode(llr32, s, it, want=WANT)), ref, f"staircase f32 {m}x{k} {kind}")
