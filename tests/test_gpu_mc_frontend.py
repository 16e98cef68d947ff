"""Device Monte-Carlo front-end (ldpc_mc_*) against its numpy
restatement (oracle/mc_frontend.py) and the decoder oracle.

The random numbers are the library's own counter-based stream (the reference seeds
rand() from the clock: parity unpinned for the stream); what is checked bit for bit is
the law of the reference's harness loop: source word, enc(), bsc() with exactly
round(n p) flips and the corrected LLR magnitude, float_to_fix / float2fix, the
classification into errors[0] / errors[1], and independence from batching / sharding.
"""
import numpy as np
import pytest

from conftest import code_path
from oracle import mc_frontend as ref

pytestmark = pytest.mark.gpu


def _setup(hip, name, alg, max_frames=128, early_exit=True):
    code = hip.Code.from_alist(code_path(name))
    dec = hip.Decoder(code, alg, max_frames=max_frames, early_exit=early_exit)
    return code, dec, hip.MonteCarlo(code, dec)


@pytest.mark.parametrize("name,F", [("204.33.484.txt", 70), ("peg_16000_3_0.9.txt", 5), ("1057.244.3.txt", 33)])
@pytest.mark.parametrize("channel,param", [(ref.BSC_EXACT, 0.031), (ref.BSC_IID, 0.05)])
@pytest.mark.parametrize("dtype", [ref.I16, ref.F64, ref.I32, ref.F32])
def test_generation_bsc_is_bit_exact(hip, name, F, channel, param, dtype):
    code, dec, mc = _setup(hip, name, hip.ALG.MINSUM_Q88)
    rp, ci, _ = code.csr()
    x, llr, s = mc.generate(F, channel, param, dtype, seed=77, first_frame=1000)
    rx, rllr, rs, d = ref.generate(code.n, rp, ci, F, channel, param, dtype, 77, first_frame=1000)
    assert np.array_equal(x, rx) and np.array_equal(s, rs)
    assert llr.dtype == rllr.dtype and np.array_equal(llr, rllr)
    if channel == ref.BSC_EXACT:                      # bsc(): exactly round(n p) flips in every frame
        flips = ((d < 0) != (rx == 1)).sum(axis=1)
        assert (flips == int(code.n * param + 0.5)).all()
    assert 0.35 < x.mean() < 0.65


def test_generation_awgn_within_libm_tolerance(hip):
    """Box-Muller goes through log / sqrt / cos: device and glibc differ in the last ulp.
    Tolerance: 1e-12 relative on the binary64 LLRs; the Q8.8 words may then differ by one
    step where an LLR sits on a rounding boundary (none expected in 14 000 values)."""
    code, dec, mc = _setup(hip, "204.33.484.txt", hip.ALG.MINSUM_Q88)
    rp, ci, _ = code.csr()
    x, llr, s = mc.generate(70, ref.AWGN, 0.8, ref.F64, seed=5)
    rx, rllr, rs, _ = ref.generate(code.n, rp, ci, 70, ref.AWGN, 0.8, ref.F64, 5)
    assert np.array_equal(x, rx) and np.array_equal(s, rs)
    assert np.allclose(llr, rllr, rtol=1e-12, atol=1e-12)
    sigma_hat = np.std(llr * 0.8 * 0.8 / 2 - (1 - 2.0 * x))
    assert abs(sigma_hat - 0.8) < 0.02
    _, q, _ = mc.generate(70, ref.AWGN, 0.8, ref.I16, seed=5)
    rq = ref.float_to_fix_raw(rllr)
    assert (np.abs(q.astype(np.int32) - rq.astype(np.int32)) <= 1).all() and (q != rq).mean() < 1e-3


def test_frames_do_not_depend_on_batching(hip):
    code, dec, mc = _setup(hip, "204.33.484.txt", hip.ALG.MINSUM_Q88)
    a = mc.generate(40, ref.BSC_EXACT, 0.04, ref.I16, seed=3, first_frame=0)
    b = mc.generate(9, ref.BSC_EXACT, 0.04, ref.I16, seed=3, first_frame=17)
    for u, v in zip(a, b):
        assert np.array_equal(u[17:26], v)
    fx = mc.generate(6, ref.BSC_EXACT, 0.04, ref.I16, seed=3, source=hip.SRC.FIXED)[0]
    assert (fx == fx[0]).all() and fx.any()
    assert not mc.generate(6, ref.BSC_EXACT, 0.04, ref.I16, seed=3, source=hip.SRC.ZERO)[0].any()


@pytest.mark.parametrize("alg_name,dtype,p,it", [("MINSUM_Q88", ref.I16, 0.055, 20), ("MINSUM_F64", ref.F64, 0.045, 20)])
def test_run_counters_match_the_oracle(hip, oracle, alg_name, dtype, p, it):
    """300 trials in passes of 128: the eight counters equal what the CPU decoder gives on the
    restated frames (both decoders are bit-exact, so the counts are too)."""
    name, F = "204.33.484.txt", 300
    code, dec, mc = _setup(hip, name, getattr(hip.ALG, alg_name))
    rp, ci, _ = code.csr()
    got = mc.run(F, ref.BSC_EXACT, p, dtype, it, seed=2024, first_frame=50)
    x, llr, s, d = ref.generate(code.n, rp, ci, F, ref.BSC_EXACT, p, dtype, 2024, first_frame=50)
    oc = oracle.load(code_path(name))
    want = dict(frames=F, not_converged=0, undetected=0, bit_errors=0, iterations=0, frames_in_error=0,
                bit_errors_converged=0)
    rcs = []
    for f in range(F):
        if alg_name == "MINSUM_Q88":
            r = oracle.dec_minsum_fixed(oc, llr[f].astype(np.float64) / 256.0, s[f].astype(np.int32), it, None, oc.m + 1)
        else:
            r = oracle.dec_minsum(oc, d[f], s[f].astype(np.int32), it)
        nerr = int((r["bits"].astype(np.uint8) != x[f]).sum())
        rcs.append(int(r["rc"] != 0))
        want["not_converged"] += r["rc"] != 0
        want["undetected"] += r["rc"] == 0 and nerr > 0
        want["bit_errors"] += nerr
        want["frames_in_error"] += nerr > 0
        want["bit_errors_converged"] += nerr if r["rc"] == 0 else 0
        want["iterations"] += r["iters"]
    assert got == {k: int(v) for k, v in want.items()}
    assert 0 < got["not_converged"] < F                      # the operating point exercises both outcomes
    # the last pass (300 - 256 = 44 frames) is still on the device
    lx, lbits, lstatus, _ = mc.last_pass(44)
    assert np.array_equal(lx, x[256:]) and (lstatus != 0).astype(int).tolist() == rcs[256:]


def test_run_is_sharding_independent(hip):
    code, dec, mc = _setup(hip, "204.33.484.txt", hip.ALG.MINSUM_Q88, max_frames=64)
    whole = mc.run(500, ref.BSC_EXACT, 0.05, ref.I16, 15, seed=9)
    a = mc.run(230, ref.BSC_EXACT, 0.05, ref.I16, 15, seed=9, first_frame=0)
    b = mc.run(270, ref.BSC_EXACT, 0.05, ref.I16, 15, seed=9, first_frame=230)
    assert whole == {k: a[k] + b[k] for k in whole}
    assert whole != mc.run(500, ref.BSC_EXACT, 0.05, ref.I16, 15, seed=10)


def test_headline_code_runs_on_device(hip, oracle):
    """PEG-16000 through the fused decoder, 600 frames in passes of 256: well below threshold
    nearly everything converges to x (plain min-sum on a column-weight-3 code keeps a small
    floor), far above it nothing does; a 40-frame run is checked against the CPU decoder."""
    name = "peg_16000_3_0.9.txt"
    code, dec, mc = _setup(hip, name, hip.ALG.MINSUM_Q88, max_frames=256)
    lo = mc.run(600, ref.BSC_EXACT, 0.001, ref.I16, 30, seed=1)
    assert lo["frames"] == 600 and lo["not_converged"] < 30 and lo["undetected"] == 0
    assert lo["bit_errors_converged"] == 0 and (lo["bit_errors"] > 0) == (lo["not_converged"] > 0)
    hi = mc.run(600, ref.BSC_IID, 0.03, ref.I16, 5, seed=1)
    assert hi["not_converged"] == 600 and hi["iterations"] == 3000 and hi["bit_errors"] > 600 * 100
    F, it, p = 40, 12, 0.0035
    got = mc.run(F, ref.BSC_EXACT, p, ref.I16, it, seed=8)
    rp, ci, _ = code.csr()
    x, llr, s, _ = ref.generate(code.n, rp, ci, F, ref.BSC_EXACT, p, ref.I16, 8)
    oc = oracle.load(code_path(name))
    res = [oracle.dec_minsum_fixed(oc, llr[f] / 256.0, s[f].astype(np.int32), it, None, oc.m + 1) for f in range(F)]
    assert got["not_converged"] == sum(r["rc"] != 0 for r in res)
    assert got["iterations"] == sum(r["iters"] for r in res)
    assert got["bit_errors"] == sum(int((r["bits"].astype(np.uint8) != x[f]).sum()) for f, r in enumerate(res))


def test_million_frames_at_the_benchmark_size(hip):
    """configs[1] at its full size -- 10^6 frames of PEG-16000, Q8.8 min-sum, passes of 65 536 --
    through size-independent properties (no CPU decoder finishes this in test time): random
    source words with s = H x (the reference's syndrome-decoding round trip, enc() -> bsc() ->
    dec() -> HamDist): a frame that reaches its syndrome at p far below threshold must have
    reached x itself (no undetected error, no bit error on converged frames); far above threshold
    no frame converges and every frame runs exactly loop_max iterations (the benchmark's
    operating point); and the counters of the whole run equal the sum over two shards of it."""
    name, F = "peg_16000_3_0.9.txt", 1_000_000
    code, dec, mc = _setup(hip, name, hip.ALG.MINSUM_Q88, max_frames=65536)
    lo = mc.run(F, ref.BSC_EXACT, 0.0005, ref.I16, 30, seed=5)
    assert lo["frames"] == F and lo["undetected"] == 0 and lo["bit_errors_converged"] == 0
    assert lo["not_converged"] < F // 1000 and (lo["bit_errors"] > 0) == (lo["not_converged"] > 0)
    assert F <= lo["iterations"] < 4 * F                       # early exit: one to three iterations per frame
    a = mc.run(400_000, ref.BSC_EXACT, 0.0005, ref.I16, 30, seed=5, first_frame=0)
    b = mc.run(F - 400_000, ref.BSC_EXACT, 0.0005, ref.I16, 30, seed=5, first_frame=400_000)
    assert lo == {k: a[k] + b[k] for k in lo}
    _, dec2, mc2 = _setup(hip, name, hip.ALG.MINSUM_Q88, max_frames=65536, early_exit=False)
    hi = mc2.run(F, ref.BSC_IID, 0.02, ref.I16, 10, seed=6)
    assert hi["frames"] == F and hi["not_converged"] == F and hi["iterations"] == 10 * F
    assert hi["frames_in_error"] == F and hi["undetected"] == 0


def test_bad_arguments(hip):
    code, dec, mc = _setup(hip, "204.33.484.txt", hip.ALG.MINSUM_Q88)
    with pytest.raises(hip.LdpcError):
        mc.run(10, ref.BSC_EXACT, 1.5, ref.I16, 5)
    with pytest.raises(hip.LdpcError):
        mc.run(10, 7, 0.1, ref.I16, 5)
    with pytest.raises(hip.LdpcError):
        mc.generate(1000, ref.BSC_EXACT, 0.1, ref.I16)       # more than one pass


def _confusion(num_reads, noise, seed=0, floor=0.0):
    """A synthetic confusion matrix p_rec_given_sent[rows][16] in the shape test_mex_B_MSDP.m
    passes: mass on the sent level, `noise` leaking to the neighbouring read bins; some
    exact zeros; columns normalised like test_code_B_MSDP.c:455-465."""
    rows = 16 if num_reads == 1 else 46
    rng = np.random.default_rng(seed)
    p = np.zeros((rows, 16))
    for j in range(16):
        centre = j if num_reads == 1 else 3 * j
        for dlt, w in ((0, 1.0), (-1, noise), (1, noise), (-2, noise / 4), (2, noise / 4)):
            i = centre + dlt
            if 0 <= i < rows:
                p[i, j] += w * (0.5 + rng.random())
    p += floor
    return p / p.sum(axis=0, keepdims=True)


@pytest.mark.parametrize("num_reads", [1, 3])
def test_qlc_front_end_is_bit_exact(hip, num_reads):
    """LDPC_CH_QLC against the restated channel() / assign_llr_one_sym() of test_code_B_MSDP.c."""
    code, dec, mc = _setup(hip, "peg_16000_3_0.9.txt", hip.ALG.SPA_F64, max_frames=64)
    conf = _confusion(num_reads, 0.06)
    mc.set_qlc_channel(conf, num_reads)
    rp, ci, _ = code.csr()
    x, llr, s = mc.generate(3, ref.QLC, 0.0, ref.F64, seed=21, first_frame=4)
    rx, rllr, rs, _ = ref.generate(code.n, rp, ci, 3, ref.QLC, 0.0, ref.F64, 21, first_frame=4, conf=conf)
    assert np.array_equal(x, rx) and np.array_equal(s, rs)
    bad = np.where(llr != rllr)
    assert len(bad[0]) == 0, (len(bad[0]), bad[0][::4].tolist(), (bad[1][::4] // 4).tolist(), llr[bad][:8], rllr[bad][:8])
    tab = ref.qlc_llr_table(conf)
    assert np.isin(llr.reshape(-1, 4), tab).all() and len(np.unique(llr)) > 8


def test_msdp_entry_point(hip):
    """test_code_B_MSDP() on the rate-1/2 (204,102) code: a mild dense matrix decodes (the CPU
    restatement of the same channel decodes 60 of 60 there), a very noisy one does not;
    decode_mode 1 and a bad read count are refused.
    Why dense and rate 1/2: a read that pins a bit gets LLR +-100 (:348-352) and tanh(50) == 1.0
    sends the reference's float sum-product to inf - inf, and its sampler maps rand() % 101 == 0
    to the first bin with non-zero probability (:428-430), i.e. 1 % of the cells of a dense
    matrix read as bin 0 -- both are the reference's behaviour, restated, not worked round."""
    import ctypes as C
    L = hip.load_library()
    path = code_path("204.33.484.txt").encode()
    err = np.zeros(2)
    dp = err.ctypes.data_as(C.POINTER(C.c_double))
    conf = np.ascontiguousarray(_confusion(3, 0.01, floor=1e-4) * 7.0)   # unnormalised on purpose
    assert L.test_code_B_MSDP(path, 30, 400, 3, 0, conf.ctypes.data_as(C.POINTER(C.c_double)), dp) == 0, L.ldpc_last_error()
    assert err[0] <= 8 and err[1] <= 1 and np.allclose(conf.sum(axis=0), 1.0)
    conf = np.ascontiguousarray(_confusion(1, 0.9))
    assert L.test_code_B_MSDP(path, 5, 40, 1, 0, conf.ctypes.data_as(C.POINTER(C.c_double)), dp) == 0
    assert err[0] >= 38
    assert L.test_code_B_MSDP(path, 5, 40, 1, 1, conf.ctypes.data_as(C.POINTER(C.c_double)), dp) == -6
    assert L.test_code_B_MSDP(path, 5, 40, 2, 0, conf.ctypes.data_as(C.POINTER(C.c_double)), dp) == -4


def test_sweep_driver_on_the_device(hip, tmp_path):
    """ldpc_amd.sweep (the loop of test_mex_B_float.m:20-36 around the harness): every point
    equals a direct ldpc_mc_run on that point's seed, chunking does not change the frames,
    the FER falls with the channel parameter, and a restart skips finished points."""
    from ldpc_amd.sweep import ebn0_to_sigma, load_points, point_seed, sweep
    code, dec, mc = _setup(hip, "204.33.484.txt", hip.ALG.MINSUM_Q88)
    ps = [0.07, 0.05, 0.03]
    out = str(tmp_path / "s.jsonl")
    recs = sweep(mc, ps, 600, ref.BSC_EXACT, ref.I16, 30, seed=4, out_path=out)
    for i, r in enumerate(recs):
        direct = mc.run(600, ref.BSC_EXACT, ps[i], ref.I16, 30, seed=point_seed(4, i))
        assert r["counters"] == direct and r["frames"] == 600
        assert r["failures"] == direct["not_converged"] and r["undetected"] == direct["undetected"]
    assert recs[0]["fer"] > recs[1]["fer"] > recs[2]["fer"] and recs[0]["fer"] > 0.2
    chunked = sweep(mc, ps, 600, ref.BSC_EXACT, ref.I16, 30, seed=4, min_failures=10 ** 9, chunk=170)
    assert [c["counters"] for c in chunked] == [r["counters"] for r in recs]
    again = sweep(mc, ps, 600, ref.BSC_EXACT, ref.I16, 30, seed=4, out_path=out)
    assert sorted(load_points(out)) == [0, 1, 2] and len(open(out).read().splitlines()) == 3
    assert [a["counters"] for a in again] == [r["counters"] for r in recs]
    # AWGN in Eb/N0: config[0]'s operating point (2 dB, 50 iterations, rate 1/2)
    sg = ebn0_to_sigma(2.0, 0.5)
    assert abs(sg - (1.0 / (2 * 0.5 * 10 ** 0.2)) ** 0.5) < 1e-15
    code2, dec2, mc2 = _setup(hip, "204.33.484.txt", hip.ALG.MINSUM_F32)
    aw = sweep(mc2, [1.0, 3.0], 2000, ref.AWGN, ref.F32, 50, seed=6, mc_params=[ebn0_to_sigma(1.0, 0.5), ebn0_to_sigma(3.0, 0.5)])
    assert aw[0]["fer"] > aw[1]["fer"] and aw[0]["avg_iterations"] > aw[1]["avg_iterations"]
