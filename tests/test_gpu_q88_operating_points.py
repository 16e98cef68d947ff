"""GPU parity of the Q8.8 min-sum engines where round 2's cases did not reach:

* the bench's own operating point -- PEG-16000, BSC p = 0.02 (above threshold: nothing converges), 10
  iterations all executed, early exit off, a frame queue several times the number of workgroups;
* saturation -- |v2c| = 127.0 (float_to_fix's ceiling, test_code_min_sum_B_fixed.c:43-53, :222-227), raw
  inputs of -32768 / +32512 / +32767, posteriors beyond +-128 (:303) -- on every engine that decodes
  Q8.8: register-cached and generic frame-resident kernels, the staircase kernel, the layered
  schedule, and the two-pass engine with each of its check kernels;
* checks of weight 1, whose message is +-200 (the minimum over nothing, :270): int32-message two-pass
  engine, frame-resident kernels, and the floating-point min-sum decoders (test_code_min_sum_B.c:168-176).

Everything is compared with the CPU oracle through the C-ABI: decoded bits, syndromes, iteration counts,
verdicts, and the posteriors of the frames that ran every iteration -- bit-exact.
"""
import numpy as np
import pytest

from conftest import code_path, to_q88
from test_gpu_binary import check_exact, run_oracle, _random_code
from test_gpu_chain import staircase_code, fused_ran

pytestmark = pytest.mark.gpu

WANT = ("bits", "iters", "synd", "status", "llr")


def ora_q88(oracle, oc, q, s, it):
    return run_oracle(lambda c, l, sy, k: oracle.dec_minsum_fixed(c, l, sy, k, None, oc.m + 1), oc, q, s, it)


def check_post(out, ref, tag):
    """Posteriors of the frames that did not converge = float_to_fix(sum) of the last iteration."""
    nc = ref[3] != 0
    if nc.any():
        got, exp = out["llr"][nc].astype(np.float64), ref[4][nc]
        bad = np.argwhere(got != exp)
        assert len(bad) == 0, f"{tag}: posteriors differ at (frame, variable) {bad[:6].tolist()}"


# --------------------------------------------------------------------- (a) the bench's operating point

def test_q88_at_the_bench_operating_point(hip, oracle):
    """bench.py's workload through the default MINSUM_Q88 engine: PEG-16000, BSC p = 0.02, max_iter 10,
    early_exit 0, 8 192 frames in one pass (32 per persistent workgroup, so the frame queue is walked).
    96 frames spread over the queue -- the first and last 16, and 64 in between -- against
    ora_dec_minsum_fixed: bits, syndromes, iteration counts, status and the posteriors of the (all)
    non-converged frames.  A quarter of the frames carry a random source word and its syndrome."""
    name = "peg_16000_3_0.9.txt"
    oc = oracle.load(code_path(name))
    F, p, it = 8192, 0.02, 10
    rng = np.random.default_rng(2024)
    raw = np.empty((F, oc.n), np.int16)
    L = int(np.floor(np.log((1 - p) / p) * 256 + 0.5))
    x = np.zeros((F, oc.n), np.uint8)
    x[::4] = rng.integers(0, 2, size=(F // 4, oc.n), dtype=np.uint8)
    for f0 in range(0, F, 512):
        flips = rng.random((512, oc.n), dtype=np.float32) < p
        raw[f0:f0 + 512] = np.where(flips ^ (x[f0:f0 + 512] != 0), -L, L)
    s = np.zeros((F, oc.m), np.uint8)
    for f in range(0, F, 4):
        s[f] = oracle.enc(oc, x[f].astype(np.int32))
    code = hip.Code.from_alist(code_path(name))
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=F, early_exit=False, want_llr=True)
    out = fused_ran(dec, lambda: dec.decode(raw, s, it, want=WANT))
    assert (out["status"] == 1).all() and (out["iters"] == it).all(), "p = 0.02 is above threshold: no frame converges"
    pick = np.r_[0:16, np.sort(rng.choice(np.arange(16, F - 16), 64, replace=False)), F - 16:F]
    ref = ora_q88(oracle, oc, raw[pick] / 256.0, s[pick].astype(np.int32), it)
    sub = {k: v[pick] for k, v in out.items()}
    check_exact(sub, ref, "bench point")
    check_post(sub, ref, "bench point")
    assert (ref[3] == 1).all()


# ------------------------------------------------------------------------------------ (b) saturation

def saturating_frames(oracle, oc, F, seed, sigma, scale=20.0, wild=True):
    """BPSK + Gaussian noise with the LLRs scaled up so that a large share of the inputs sit on
    float_to_fix's limits, and 1 % of the positions forced to the raw extremes: -32768 (= -128.0),
    +32512 (= 127.0) and, with `wild`, +32767 (127.996: a raw int16 the quantiser never produces but
    dec() would take as it is)."""
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 2, size=(F, oc.n)).astype(np.int32)
    x[:F // 2] = 0
    s = np.stack([oracle.enc(oc, x[f]) for f in range(F)]).astype(np.int32)
    y = 1.0 - 2.0 * x + rng.normal(0, sigma, size=(F, oc.n))
    _, raw = to_q88(2 * y / sigma ** 2 * scale)
    pick = rng.random(raw.shape) < 0.01
    top = np.where(rng.random(raw.shape) < 0.3, 32767, 32512) if wild else 32512
    raw = np.where(pick & (raw < 0), -32768, raw)
    raw = np.where(pick & (raw > 0), top, raw).astype(np.int16)
    assert (raw == -32768).any() and (raw == 32512).any()
    return x, s, raw


def assert_saturated(ref, raw):
    post = ref[4]
    assert (np.abs(raw.astype(np.int32)) >= 32512).mean() > 0.01
    assert post.max() == 127.0 and post.min() == -128.0, "posteriors never left [-128, 127]: the case does not saturate"


SAT_CODES = [
    # code, frames, sigma, iterations
    ("204.33.484.txt", 96, 0.75, 20),
    ("Frolov_1024_0.5.txt", 40, 0.8, 30),
    ("1057.244.3.txt", 64, 0.58, 15),
    ("4095.737.3.txt", 24, 0.54, 12),
    ("peg_16000_3_0.9.txt", 10, 0.455, 10),
    ("32000.2240.3.105.txt", 5, 0.4, 12),
]

ENGINES = {
    # name: (environment, Decoder keywords)
    "default": ({}, {}),
    "generic_fused": ({"LDPC_HIP_FUSED_REG": "0", "LDPC_HIP_FUSED": "2"}, {}),
    "two_pass_single_read": ({"LDPC_HIP_FUSED": "0", "LDPC_HIP_CHAIN": "0"}, {}),
    "two_pass_lds": ({}, {"cn_variant": 1}),
    "two_pass_reload": ({}, {"cn_variant": 2}),
    "two_pass_registers": ({}, {"cn_variant": 3}),
    "two_pass_dma": ({}, {"cn_variant": 4}),
}


@pytest.mark.parametrize("engine", list(ENGINES))
@pytest.mark.parametrize("name,F,sigma,it", SAT_CODES)
def test_q88_saturation(hip, oracle, monkeypatch, name, F, sigma, it, engine):
    oc = oracle.load(code_path(name))
    x, s, raw = saturating_frames(oracle, oc, F, 91, sigma)
    ref = ora_q88(oracle, oc, raw / 256.0, s, it)
    assert_saturated(ref, raw)
    env, kw = ENGINES[engine]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    code = hip.Code.from_alist(code_path(name))
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=128, early_exit=True, want_llr=True, **kw)
    su = s.astype(np.uint8)
    out = dec.decode(raw, su, it, want=WANT)
    check_exact(out, ref, f"saturation {engine} {name} raw int16")
    check_post(out, ref, f"saturation {engine} {name}")
    # the same values as doubles and as int32 raw (only inputs inside the quantiser's range: +32767 is
    # a raw-int16-only value, float_to_fix would turn 127.996 into 127.0)
    tame = np.minimum(raw, 32512)
    ref2 = ora_q88(oracle, oc, tame / 256.0, s, it)
    for inp in (tame / 256.0, (tame / 256.0).astype(np.float32), tame.astype(np.int32)):
        check_exact(dec.decode(inp, su, it, want=WANT), ref2, f"saturation {engine} {name} {inp.dtype}")
    print(f"saturation {engine} {name}: {(ref[3] == 0).sum()} of {F} frames converge, iterations {sorted(set(ref[1].tolist()))}")


@pytest.mark.parametrize("name,layer,F,sigma,it", [("204.33.484.txt", 64, 96, 0.75, 20), ("1057.244.3.txt", 64, 64, 0.58, 15),
                                                   ("4095.737.3.txt", 128, 24, 0.56, 12), ("peg_16000_3_0.9.txt", 256, 8, 0.47, 10)])
def test_q88_saturation_layered(hip, oracle, name, layer, F, sigma, it):
    oc = oracle.load(code_path(name))
    x, s, raw = saturating_frames(oracle, oc, F, 92, sigma)
    ref = run_oracle(lambda c, l, sy, k: oracle.dec_minsum_fixed_layered(c, l, sy, k, layer), oc, raw / 256.0, s, it)
    assert_saturated(ref, raw)
    code = hip.Code.from_alist(code_path(name))
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=128, early_exit=True, want_llr=True, layer_checks=layer)
    out = dec.decode(raw, s.astype(np.uint8), it, want=WANT)
    check_exact(out, ref, f"saturation layered {name} L={layer}")
    check_post(out, ref, f"saturation layered {name}")


@pytest.mark.parametrize("m,k,iw,holes,tail,six,F,sigma,it", [(250, 200, (5, 5), 0, True, 0, 40, 0.9, 12),
                                                             (1500, 1200, (3, 5), 20, False, 11, 16, 0.8, 20),
                                                             (4096, 3000, (4, 5), 0, True, 3, 6, 0.85, 25)])
def test_q88_saturation_on_the_staircase_kernel(hip, oracle, tmp_path, monkeypatch, m, k, iw, holes, tail, six, F, sigma, it):
    rng = np.random.default_rng(m + k)
    H = staircase_code(rng, m, k, iw, holes, tail, six)
    path = str(tmp_path / "stair.alist")
    hip.alist.matrix_to_alist(H, path)
    oc = oracle.load(path)
    x, s, raw = saturating_frames(oracle, oc, F, 93, sigma)
    ref = ora_q88(oracle, oc, raw / 256.0, s, it)
    assert_saturated(ref, raw)
    monkeypatch.setenv("LDPC_HIP_CHAIN", "2")
    code = hip.Code.from_alist(path)
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=64, early_exit=True, want_llr=True)
    out = fused_ran(dec, lambda: dec.decode(raw, s.astype(np.uint8), it, want=WANT))
    check_exact(out, ref, f"saturation staircase {m}x{m + k}")
    check_post(out, ref, f"saturation staircase {m}x{m + k}")


def test_q88_saturation_on_dvb_s2(hip, oracle, tmp_path):
    """configs[2]'s code on its default Q8.8 engine (the staircase kernel), inputs on the limits."""
    path = str(tmp_path / "padded.alist")
    hip.alist.normalise_alist(code_path("dvb_H_0.5.txt"), path, padded=True)
    oc = oracle.load(path)
    x, s, raw = saturating_frames(oracle, oc, 4, 94, 1.0)
    ref = ora_q88(oracle, oc, raw / 256.0, s, 12)
    assert_saturated(ref, raw)
    code = hip.Code.from_alist(code_path("dvb_H_0.5.txt"))
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=16, early_exit=True, want_llr=True)
    out = fused_ran(dec, lambda: dec.decode(raw, s.astype(np.uint8), 12, want=WANT))
    check_exact(out, ref, "saturation dvb-s2")
    check_post(out, ref, "saturation dvb-s2")


@pytest.mark.parametrize("heavy", [39, 24])
@pytest.mark.parametrize("fused", ["1", "2", "0"])
def test_q88_saturation_with_the_heaviest_column_the_fused_kernel_takes(hip, oracle, tmp_path, monkeypatch, heavy, fused):
    """One variable in 39 checks (fq_cmax_ok's limit; 40 goes to the two-pass engine): with every
    message at +-127 its sum reaches 128 + 39 * 127 = 5081, the far end of the biased LDS words' exact range."""
    rng = np.random.default_rng(heavy)
    n, m = 1500, 300
    H = _random_code(rng, n, m, 3)
    H[:, 7] = 0
    H[rng.choice(m, heavy, replace=False), 7] = 1
    path = str(tmp_path / "heavy.alist")
    hip.alist.matrix_to_alist(H, path)
    oc = oracle.load(path)
    x, s, raw = saturating_frames(oracle, oc, 32, 95, 0.55)
    raw[:, 7] = np.where(np.arange(32) % 2 == 0, -32768, 32512)       # the heavy variable itself on the limits
    ref = ora_q88(oracle, oc, raw / 256.0, s, 12)
    assert_saturated(ref, raw)
    monkeypatch.setenv("LDPC_HIP_FUSED", fused)
    code = hip.Code.from_alist(path)
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=64, early_exit=True, want_llr=True)
    out = dec.decode(raw, s.astype(np.uint8), 12, want=WANT)
    if fused != "0":
        dec.profile(True)
        dec.decode(raw, s.astype(np.uint8), 12)
        _, cnt = dec.profile_read()
        assert cnt[2] > 0 and cnt[1] == 0, "column weight 39 is within the fused kernel's range"
    check_exact(out, ref, f"heavy column {heavy} LDPC_HIP_FUSED={fused}")
    check_post(out, ref, f"heavy column {heavy}")


# --------------------------------------------------------------------------- (c) checks of weight 1

def weight_one_code(rng, n, m, dv, singles):
    """A random (dv, .)-regular-ish code with `singles` extra rows of weight 1."""
    H = _random_code(rng, n, m - singles, dv)
    one = np.zeros((singles, n), np.uint8)
    one[np.arange(singles), rng.choice(n, singles, replace=False)] = 1
    # interleave so that weight-1 rows sit among the others (edge numbering, wave-checks)
    out = np.zeros((m, n), np.uint8)
    where = np.sort(rng.choice(m, singles, replace=False))
    mask = np.zeros(m, bool)
    mask[where] = True
    out[mask] = one
    out[~mask] = H
    return out


@pytest.mark.parametrize("n,m,dv,singles,sigma", [(600, 200, 3, 17, 0.7), (2000, 400, 3, 40, 0.55), (96, 48, 3, 5, 0.7)])
@pytest.mark.parametrize("kind", ["awgn", "sat"])
def test_weight_one_checks(hip, oracle, tmp_path, monkeypatch, n, m, dv, singles, sigma, kind):
    rng = np.random.default_rng(n + singles)
    H = weight_one_code(rng, n, m, dv, singles)
    assert (H.sum(axis=1) == 1).sum() == singles
    path = str(tmp_path / "w1.alist")
    hip.alist.matrix_to_alist(H, path)
    oc = oracle.load(path)
    code = hip.Code.from_alist(path)
    F, it = 48, 12
    if kind == "sat":
        x, s, raw = saturating_frames(oracle, oc, F, 96, sigma, wild=False)
    else:
        x, s, raw = saturating_frames(oracle, oc, F, 96, sigma, scale=1.0, wild=False)
    su = s.astype(np.uint8)
    q = raw / 256.0
    ref = ora_q88(oracle, oc, q, s, it)
    assert np.abs(ref[4]).max() >= 127.0 or kind == "awgn"
    assert len(set(ref[1].tolist())) > 1
    for fused in ("1", "2", "0"):
        monkeypatch.setenv("LDPC_HIP_FUSED", fused)
        dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=64, early_exit=True, want_llr=True)
        for inp in (raw, q):
            out = dec.decode(inp, su, it, want=WANT)
            check_exact(out, ref, f"weight-1 checks q88 LDPC_HIP_FUSED={fused} {inp.dtype}")
            check_post(out, ref, f"weight-1 checks q88 LDPC_HIP_FUSED={fused}")
    monkeypatch.delenv("LDPC_HIP_FUSED")
    for variant in (1, 2, 3, 4, 5):
        dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=64, early_exit=True, want_llr=True, cn_variant=variant)
        out = dec.decode(raw, su, it, want=WANT)
        check_exact(out, ref, f"weight-1 checks q88 cn_variant {variant}")
        check_post(out, ref, f"weight-1 checks q88 cn_variant {variant}")
    # the floating-point decoders: c2v = +-200 exactly (test_code_min_sum_B.c:168-176)
    ref64 = run_oracle(oracle.dec_minsum, oc, q, s, it)
    ref32 = run_oracle(oracle.dec_minsum_f32, oc, q.astype(np.float32), s, it)
    assert np.abs(ref64[4]).max() >= 200.0, "no posterior carries a weight-1 check's 200"
    for variant in (0, 2, 5):
        d64 = hip.Decoder(code, hip.ALG.MINSUM_F64, max_frames=64, want_llr=True, cn_variant=variant)
        out = d64.decode(q, su, it, want=WANT)
        check_exact(out, ref64, f"weight-1 checks f64 cn_variant {variant}")
        nc = ref64[3] != 0
        assert np.array_equal(out["llr"][nc], ref64[4][nc].astype(np.float32))
        d32 = hip.Decoder(code, hip.ALG.MINSUM_F32, max_frames=64, want_llr=True, cn_variant=variant)
        out = d32.decode(q.astype(np.float32), su, it, want=WANT)
        check_exact(out, ref32, f"weight-1 checks f32 cn_variant {variant}")
        nc = ref32[3] != 0
        assert np.array_equal(out["llr"][nc], ref32[4][nc])
