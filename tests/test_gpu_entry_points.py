"""GPU: the reference-shaped entry points and the Verilog trace, through the C-ABI."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, code_path

pytestmark = pytest.mark.gpu


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_verilog_trace_kat_from_gpu(hip, tmp_path):
    """Verilog_LDPC_decoder_tester.m: test_code_min_sum_B_fixed('Frolov_1024_0.5.txt', 30, ...,
    'verilog.txt') -- every v2c / c2v message of every iteration, read back from the GPU,
    byte-identical to the reference's stored verilog.txt."""
    L = hip.load_library()
    kat = np.load(os.path.join(GOLDEN, "frolov_kat.npz"))
    want = gzip.open(os.path.join(GOLDEN, "verilog_success.txt.gz"), "rb").read()
    y = kat["y_2048_success"].astype(np.int32)
    n, m = 2048, 1024
    x = np.zeros(n, np.int32)
    s = np.zeros(m, np.int32)
    q0 = np.where(y == 0, 11 / 256, -11 / 256).astype(np.float64)
    err = np.zeros(2)
    out = str(tmp_path / "verilog.txt")
    rc = L.test_code_min_sum_B_fixed(code_path("Frolov_1024_0.5.txt").encode(), 30, _ip(x), _ip(s), _ip(y),
                                     _dp(q0), _dp(err), out.encode())
    assert rc == 0, L.ldpc_last_error()
    assert err.tolist() == [0.0, 0.0]
    assert open(out, "rb").read() == want
    # without a trace file: same verdict through the batch path
    rc = L.test_code_min_sum_B_fixed(code_path("Frolov_1024_0.5.txt").encode(), 30, _ip(x), _ip(s), _ip(y),
                                     _dp(q0), _dp(err), None)
    assert rc == 0 and err.tolist() == [0.0, 0.0]


def test_verilog_trace_nonconvergent_vs_oracle(hip, oracle, tmp_path):
    kat = np.load(os.path.join(GOLDEN, "frolov_kat.npz"))
    y = kat["y_2048"].astype(np.int32)
    q0 = np.where(y == 0, 11 / 256, -11 / 256).astype(np.float64)
    oc = oracle.load(code_path("Frolov_1024_0.5.txt"))
    t_ref = str(tmp_path / "ref.txt")
    r = oracle.dec_minsum_fixed(oc, q0, np.zeros(oc.m, np.int32), 30, t_ref, 128)
    code = hip.Code.from_alist(code_path("Frolov_1024_0.5.txt"))
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=64)
    t_gpu = str(tmp_path / "gpu.txt")
    out = dec.decode_trace_q88(q0, np.zeros(oc.m, np.uint8), 30, t_gpu)
    assert out["status"] == r["rc"] == 1 and out["iters"] == r["iters"] == 30
    assert np.array_equal(out["bits"], r["bits"]) and int(out["bits"].sum()) == 107
    a, b = open(t_gpu, "rb").read(), open(t_ref, "rb").read()
    assert len(b.splitlines()) == 437760 and a == b
    # "already cw" (:205): a word that satisfies the syndrome before any iteration
    out = dec.decode_trace_q88(np.full(oc.n, 11 / 256), None, 30, t_gpu)
    assert out["status"] == 0 and out["iters"] == 0 and open(t_gpu).read() == "already cw\n"
    # a code whose rows inside a 128-block differ in weight: refused, as documented
    peg = hip.Code.from_alist(code_path("peg_16000_3_0.9.txt"))
    d2 = hip.Decoder(peg, hip.ALG.MINSUM_Q88, max_frames=64)
    with pytest.raises(hip.LdpcError) as e:
        d2.decode_trace_q88(np.full(peg.n, 1.0), None, 3, t_gpu)
    assert e.value.code == -6


@pytest.mark.parametrize("entry,alg_fn", [("test_code_min_sum_B", "dec_minsum"), ("test_code_B_float", "dec_spa")])
def test_single_frame_entry_points(hip, oracle, entry, alg_fn):
    L = hip.load_library()
    path = code_path("204.33.484.txt")
    oc = oracle.load(path)
    rng = np.random.default_rng(5)
    fn = getattr(L, entry)
    seen = set()
    for trial in range(12):
        p = [0.02, 0.06, 0.11][trial % 3]
        x = rng.integers(0, 2, oc.n).astype(np.int32)
        s = oracle.enc(oc, x)
        y = (x ^ (rng.random(oc.n) < p)).astype(np.int32)
        q0 = ((1 - 2 * y) * np.log((1 - p) / p)).astype(np.float64)
        err = np.zeros(2)
        assert fn(path.encode(), 25, _ip(x), _ip(s), _ip(y), _dp(q0), _dp(err)) == 0
        r = getattr(oracle, alg_fn)(oc, q0, s, 25)
        want = [1.0, 0.0] if r["rc"] else [0.0, float(not np.array_equal(r["bits"], x))]
        assert err.tolist() == want
        seen.add(tuple(want))
    assert len(seen) >= 2


def test_fixed_single_frame_takes_on_grid_llrs_only(hip, oracle):
    """test_code_min_sum_B of the fixed file (_fixed.c:442-473) hands q0 to dec() untouched, so
    an off-grid q0 reaches its sums and its iteration-0 test unquantised; the int16 decoder is
    exact for Q8.8 values only.  On-grid frames (what the reference's bsc() :117-120 makes):
    equal to the oracle.  Off-grid or out-of-range q0: refused (LDPC_ERR_UNSUPPORTED), not
    silently rounded -- q0 = -0.001 is bit 1 in the reference and would round to +0 here."""
    L = hip.load_library()
    path = code_path("204.33.484.txt")
    oc = oracle.load(path)
    rng = np.random.default_rng(9)
    for trial in range(6):
        p = [0.03, 0.08][trial % 2]
        x = rng.integers(0, 2, oc.n).astype(np.int32)
        s = oracle.enc(oc, x)
        y = (x ^ (rng.random(oc.n) < p)).astype(np.int32)
        Lq = round(np.log((1 - p) / p) * 256) / 256.0
        q0 = ((1 - 2 * y) * Lq).astype(np.float64)
        err = np.zeros(2)
        assert L.test_code_min_sum_B_fixed(path.encode(), 25, _ip(x), _ip(s), _ip(y), _dp(q0), _dp(err), None) == 0
        r = oracle.dec_minsum_fixed(oc, q0, s, 25, None, oc.m + 1)
        assert err.tolist() == ([1.0, 0.0] if r["rc"] else [0.0, float(not np.array_equal(r["bits"], x))])
    for bad in (-0.001, np.log((1 - 0.03) / 0.03), 127.5, -128.25):
        q = q0.copy()
        q[17] = bad
        err = np.full(2, 7.0)
        rc = L.test_code_min_sum_B_fixed(path.encode(), 25, _ip(x), _ip(s), _ip(y), _dp(q), _dp(err), None)
        assert rc == -6 and b"q0[17]" in L.ldpc_last_error() and err.tolist() == [0.0, 0.0]
    # the oracle shows why: the raw value decides the iteration-0 word, the rounded one would not
    q = np.full(oc.n, 4.0)
    q[17] = -0.001
    r = oracle.dec_minsum_fixed(oc, q, np.zeros(oc.m, np.int32), 0, None, oc.m + 1)
    assert r["bits"][17] == 1 and oracle.float_to_fix(-0.001) == 0.0


def test_monte_carlo_entry_points(hip):
    """*_cw_noise_gen and test_code_B: Monte Carlo over the library's own stream, so the
    check is statistical, as for the reference (srand(time(NULL)))."""
    L = hip.load_library()
    L.ldpc_set_seed(99)
    err = np.zeros(2)
    path = code_path("1057.244.3.txt").encode()
    for fn in (L.test_code_min_sum_B_cw_noise_gen, L.test_code_min_sum_B_fixed_cw_noise_gen,
               L.test_code_B_float_cw_noise_gen):
        assert fn(path, 30, 300, 0.002, _dp(err)) == 0, L.ldpc_last_error()
        assert err[0] <= 3 and err[1] == 0, (fn.__name__, err)
        assert fn(path, 5, 200, 0.12, _dp(err)) == 0
        assert err[0] >= 195, (fn.__name__, err)
    assert L.test_code_B(30, 200, 0.004, path, _dp(err)) == 0
    assert err[0] <= 2 and err[1] == 0
    assert L.test_code_B(5, 100, 0.12, path, _dp(err)) == 0
    assert err[0] >= 95
    # reproducible under a fixed seed
    L.ldpc_set_seed(7)
    a = np.zeros(2)
    L.test_code_min_sum_B_cw_noise_gen(path, 15, 400, 0.007, _dp(a))
    L.ldpc_set_seed(7)
    b = np.zeros(2)
    L.test_code_min_sum_B_cw_noise_gen(path, 15, 400, 0.007, _dp(b))
    assert a.tolist() == b.tolist() and 0 < a[0] < 400


def test_monte_carlo_entry_points_over_a_device_list(hip, monkeypatch):
    """LDPC_HIP_DEVICES spreads the trials of a harness call over worker threads, one decoder per entry of the
    list.  Frames are keyed by their global index, so the counters must not depend on the list: the same call
    on one device, on a list that names this box's one GPU three times, and on "all"."""
    L = hip.load_library()
    path = code_path("1057.244.3.txt").encode()
    res = {}
    for devs in (None, "0,0,0", "all", "0,0"):
        if devs is None:
            monkeypatch.delenv("LDPC_HIP_DEVICES", raising=False)
        else:
            monkeypatch.setenv("LDPC_HIP_DEVICES", devs)
        out = []
        for fn, p in ((L.test_code_min_sum_B_fixed_cw_noise_gen, 0.0075), (L.test_code_B_float_cw_noise_gen, 0.012)):
            L.ldpc_set_seed(2026)
            err = np.zeros(2)
            assert fn(path, 12, 1001, p, _dp(err)) == 0, L.ldpc_last_error()
            out.append(err.tolist())
        L.ldpc_set_seed(5)
        err = np.zeros(2)
        assert L.test_code_B(12, 777, 0.011, path, _dp(err)) == 0, L.ldpc_last_error()
        out.append(err.tolist())
        res[devs] = out
    print(res)
    assert all(0 < e[0] < 1001 for e in res[None]), "the points must sit in the waterfall"
    for devs in ("0,0,0", "all", "0,0"):
        assert res[devs] == res[None], devs
    # an entry that is no device: the error comes back from the worker thread
    monkeypatch.setenv("LDPC_HIP_DEVICES", "0,99")
    err = np.zeros(2)
    assert L.test_code_min_sum_B_cw_noise_gen(path, 5, 100, 0.01, _dp(err)) != 0
    assert L.ldpc_last_error()


def test_python_run_api(hip, oracle):
    """ldpc_decode.run(H, LLR, max_iter): the call shape of LDPC_decode.py:348-374."""
    from ldpc_amd import ldpc_decode
    from ldpc_amd.alist import alist_to_mat
    path = code_path("204.33.484.txt")
    H = alist_to_mat(path)
    oc = oracle.load(path)
    rng = np.random.default_rng(8)
    p = 0.05
    for trial in range(6):
        flips = rng.random(oc.n) < (p if trial else 0.0)
        llr = (1 - 2 * flips.astype(int)) * np.log((1 - p) / p)
        cw, it, secs = ldpc_decode.run(H, llr, 30)
        r = oracle.dec_spa(oc, llr.astype(np.float64), np.zeros(oc.m, np.int32), 30)
        assert cw.shape == (oc.n, 1) and secs >= 0
        assert (it is False) == (r["rc"] != 0)
        if it is not False:
            assert it == r["iters"] and np.array_equal(cw[:, 0], r["bits"])
    cw, it, _ = ldpc_decode.run(H, np.full(oc.n, 3.0), 30, precheck=True)
    assert it is True and not cw.any()
    out = ldpc_decode.run_batch(path, np.full((5, oc.n), 2.5), 10)
    assert (out["status"] == 0).all() and not out["bits"].any()


@pytest.mark.parametrize("tag,fname", [("c204", "204.33.484.txt"), ("small", "small_code.txt")])
def test_python_run_api_vs_reference_vectors(hip, tag, fname):
    """ldpc_decode.run(H, LLR, max_iter, precheck=True) against the outputs of the reference's
    parallel.run (tests/golden/python_run.npz): decoded word (n,1), and the iteration slot --
    True when the input already satisfies H, the 1-based iteration, or False."""
    from ldpc_amd import ldpc_decode
    from ldpc_amd.alist import alist_to_mat
    g = np.load(os.path.join(GOLDEN, "python_run.npz"))
    H = alist_to_mat(code_path(fname))
    maxit = int(g[tag + "_maxit"])
    for llr, cw, it in zip(g[tag + "_llr"], g[tag + "_cw"], g[tag + "_iters"]):
        got_cw, got_it, secs = ldpc_decode.run(H, llr, maxit, precheck=True)
        assert got_cw.shape == (H.shape[1], 1) and np.array_equal(got_cw[:, 0], cw)
        want = True if it == 0 else (False if it < 0 else int(it))
        assert got_it is want if isinstance(want, bool) else (got_it == want and got_it is not True)


def test_plain_c_driver_programs():
    """The reference's two stand-alone programs rebuilt as plain-C callers of libldpc_hip
    (tools/drivers, gcc): test_code_no_mex_B's main() arguments (:386) and GFq_LDPC_NTT's
    usage example (:661), whose three verdict lines are those of the compiled reference."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "drivers", "bin", "test_code_no_mex_B")
    r = subprocess.run([exe, code_path("4095.737.3.txt"), "30", "4", "0.01"], capture_output=True, text=True,
                       timeout=120, env=dict(os.environ, LDPC_HIP_SEED="7"))
    assert r.returncode == 0, r.stderr
    assert r.stdout == "Errors = 0.000000\nUndetected Errors = 0.000000\n"
    exe = os.path.join(ROOT, "tools", "drivers", "bin", "GFq_LDPC_NTT")
    r = subprocess.run([exe, "-iter", "50", "bsc", "0.08", code_path("q4.sp.9000.6000.4500.1")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "converged.\nconverged.\nconverged.\n"
