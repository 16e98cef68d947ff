"""GPU parity of the non-binary decoders (through the C-ABI) against the oracle and
against the committed outputs of the compiled reference GFq_LDPC_NTT.c."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, code_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("q,fname", [(4, "q4.sp.9000.6000.4500.1"), (8, "q8.sp.6000.4000.3000.1")])
def test_gf2p_wht_golden_reference_vectors(hip, q, fname):
    """main():681-695 experiments of the reference program (seeds 1..3 + one
    non-convergent frame): decoded symbols, syndrome and verdict, bit-exact."""
    g = np.load(os.path.join(GOLDEN, f"ref_ntt_q{q}.npz"))
    code = hip.Code.from_alist(code_path(fname))
    assert code.q == q
    dec = hip.Decoder(code, hip.ALG.GF2P_WHT, max_frames=16)
    out = dec.decode(g["logfna"].astype(np.int32), g["s"], int(g["maxit"]))
    assert np.array_equal(out["status"] != 0, g["rc"] != 0)
    assert np.array_equal(out["bits"], g["xhat"])
    assert np.array_equal(out["synd"], g["zhat"])
    conv = g["rc"] == 0
    assert np.array_equal(out["bits"][conv], g["x"][conv])


def _wht_frames(oracle, oc, q, F, p, seed):
    """BSC on the symbol bits, log2-domain Q.14 priors (GFq_LDPC_NTT.c:278-315)."""
    rng = np.random.default_rng(seed)
    log2q = int(np.log2(q))
    x = rng.integers(0, q, size=(F, oc.n)).astype(np.int32)
    x[: F // 3] = 0
    s = np.stack([oracle.gf_enc(oc, q, x[f]) for f in range(F)])
    p = np.where(np.arange(F) % 2 == 0, p, 1.75 * p)[:, None]      # every other frame is harder
    err = np.zeros((F, oc.n), np.int32)
    for b in range(log2q):
        err |= (rng.random((F, oc.n)) < p).astype(np.int32) << b
    y = x ^ err
    a = np.arange(q)
    diff = (y[:, :, None] ^ a[None, None, :])
    nb = np.zeros_like(diff)
    for b in range(log2q):
        nb += (diff >> b) & 1
    logp = nb * np.log2(p)[:, :, None] + (log2q - nb) * np.log2(1 - p)[:, :, None]
    return x, s, oracle.wht_float2fix(logp), logp


@pytest.mark.parametrize("q,fname,F,p,it", [(4, "q4.sp.9000.6000.4500.1", 21, 0.085, 30),
                                            (8, "q8.sp.6000.4000.3000.1", 19, 0.07, 30)])
def test_gf2p_wht_vs_oracle(hip, oracle, q, fname, F, p, it):
    oc = oracle.load(code_path(fname))
    x, s, logfna, logp = _wht_frames(oracle, oc, q, F, p, 21)
    log2q = int(np.log2(q))
    ref = [oracle.dec_gf2p_wht(oc, log2q, logfna[f], s[f], it) for f in range(F)]
    assert all(r["rc"] in (0, -1) for r in ref)
    code = hip.Code.from_alist(code_path(fname))
    dec = hip.Decoder(code, hip.ALG.GF2P_WHT, max_frames=16)        # F > max_frames: two passes
    for inp in (logfna.astype(np.int32), logp.astype(np.float64)):
        out = dec.decode(inp, s.astype(np.uint8), it)
        assert np.array_equal(out["status"], np.array([0 if r["rc"] == 0 else 1 for r in ref]))
        assert np.array_equal(out["iters"], np.array([r["iters"] for r in ref]))
        assert np.array_equal(out["bits"], np.stack([r["x"] for r in ref]))
        assert np.array_equal(out["synd"], np.stack([r["z"] for r in ref]))
    assert any(r["rc"] == 0 for r in ref)


def _random_nb_code(rng, q, m, n, row_w, col_cap, heavy=0):
    """Random H[m][n] over GF(q): row weights drawn from row_w, no column above col_cap, no empty column;
    `heavy` columns are then filled up to weight 5 (rows stay within row_w[1] + 1)."""
    H = np.zeros((m, n), np.int32)
    colw = np.zeros(n, np.int32)
    for j in range(m):
        w = int(rng.integers(row_w[0], row_w[1] + 1))
        free = np.flatnonzero(colw < col_cap)
        pick = rng.choice(free, size=w, replace=False)
        H[j, pick] = rng.integers(1, q, size=w)
        colw[pick] += 1
    for v in np.flatnonzero(colw == 0):          # every symbol is checked, by a row that has room
        roww = np.count_nonzero(H, axis=1)
        H[int(rng.choice(np.flatnonzero(roww < row_w[1]))), v] = int(rng.integers(1, q))
    for v in rng.choice(n, size=heavy, replace=False):
        while np.count_nonzero(H[:, v]) < 5:
            roww = np.count_nonzero(H, axis=1)
            H[int(rng.choice(np.flatnonzero((roww <= row_w[1]) & (H[:, v] == 0)))), v] = int(rng.integers(1, q))
    return H


@pytest.mark.parametrize("q,m,n,row_w,col_cap,heavy,p,mode", [
    (4, 150, 400, (5, 8), 3, 0, 0.01, "1"),      # GF(4), row weights 5..8: the PF = 8 build of the specialised check kernel
    (8, 120, 320, (5, 8), 3, 0, 0.01, "1"),      # GF(8), the same
    (4, 150, 400, (5, 7), 3, 4, 0.03, "1"),      # four columns of weight 5: the PF = 8 build of the variable kernel
    (8, 120, 320, (5, 7), 3, 4, 0.03, "1"),
    (4, 90, 150, (3, 4), 4, 0, 0.012, "2"),      # the generic LDS-table kernels on a GF(4) code
    (16, 60, 100, (3, 5), 4, 0, 0.05, "1"),      # GF(16): not specialised, generic LDS-table kernels
    (4, 100, 420, (9, 10), 3, 0, 0.005, "1"),    # row weight beyond 8: generic LDS-table kernels
])
def test_gf2p_wht_random_codes(hip, oracle, tmp_path, monkeypatch, q, m, n, row_w, col_cap, heavy, p, mode):
    """Every build of the GF(2^p) kernels against the oracle of GFq_LDPC_NTT.c:437-544 on small random
    codes (the reference's two codes only reach the weight <= 4 builds): decoded symbols, iteration
    counts, verdicts and syndromes, bit-exact."""
    rng = np.random.default_rng(q * 1000 + m)
    H = _random_nb_code(rng, q, m, n, row_w, col_cap, heavy)
    path = str(tmp_path / "nb.alist")
    hip.alist.matrix_to_alist_NB(H, path, q)
    oc = oracle.load(path)
    log2q = int(np.log2(q))
    F, it = 37, 12
    x, s, logfna, logp = _wht_frames(oracle, oc, q, F, p, 5)
    ref = [oracle.dec_gf2p_wht(oc, log2q, logfna[f], s[f], it) for f in range(F)]
    # rc = -2: a message left the range of the reference's unchecked Fexp table (undefined behaviour
    # there, GFq_LDPC_NTT.c:236; the library clamps the index) -- such frames have no reference result
    ok = np.array([r["rc"] in (0, -1) for r in ref])
    assert ok.sum() >= F // 2 and len(set(r["iters"] for r in ref if r["rc"] != -2)) >= 2
    assert any(r["rc"] == -1 for r in ref)
    monkeypatch.setenv("LDPC_HIP_GF_LDS_TABLES", mode)
    code = hip.Code.from_alist(path)
    dec = hip.Decoder(code, hip.ALG.GF2P_WHT, max_frames=64)
    out = dec.decode(logfna.astype(np.int32), s.astype(np.uint8), it)
    assert np.array_equal(out["status"][ok], np.array([0 if r["rc"] == 0 else 1 for r in ref])[ok])
    assert np.array_equal(out["iters"][ok], np.array([r["iters"] for r in ref])[ok])
    assert np.array_equal(out["bits"][ok], np.stack([r["x"] for r in ref])[ok])
    assert np.array_equal(out["synd"][ok], np.stack([r["z"] for r in ref])[ok])


def _conf_matrix(q, perr):
    """Symmetric +-1 confusion matrix in the shape test_mex_NB_p.m:35-47 builds
    ([rec][sent], one read)."""
    c = np.zeros((q, q))
    for sent in range(q):
        c[sent, sent] = 1 - perr
        c[(sent + 1) % q, sent] += perr / 2
        c[(sent - 1) % q, sent] += perr / 2
    return c


@pytest.mark.parametrize("fname,q,F,perr,it", [("peg_4376_4_0.9_NB.txt", 13, 6, 0.012, 12),
                                               ("q4.sp.9000.6000.4500.1", 4, 5, 0.12, 12)])
def test_gfq_direct_vs_oracle(hip, oracle, fname, q, F, perr, it):
    """binary64 log-domain decoder: libm differs between host and device, so parity is on
    decoded symbols / iterations / verdict (identical on these inputs)."""
    oc = oracle.load(code_path(fname))
    rng = np.random.default_rng(31)
    conf = _conf_matrix(q, perr)
    x = rng.integers(0, q, size=(F, oc.n)).astype(np.int32)
    x[0] = 0
    s = np.stack([oracle.gf_enc(oc, q, x[f]) for f in range(F)])
    u = rng.random((F, oc.n))
    pe = np.where(np.arange(F) % 3 == 2, 5 * perr, perr)[:, None]  # every third frame is hopeless
    y = np.where(u < pe / 2, (x + 1) % q, np.where(u < pe, (x - 1) % q, x)).astype(np.int32)
    logfna = np.stack([oracle.gf_assign_llr(q, 1, conf, y[f]) for f in range(F)])      # [F][q][n]
    ref = [oracle.dec_gfq_direct(oc, q, logfna[f], s[f], it) for f in range(F)]
    code = hip.Code.from_alist(code_path(fname))
    dec = hip.Decoder(code, hip.ALG.GFQ_DIRECT, max_frames=4)
    out = dec.decode(np.ascontiguousarray(logfna.transpose(0, 2, 1)), s.astype(np.uint8), it)
    assert np.array_equal(out["status"], np.array([r["rc"] for r in ref]))
    assert np.array_equal(out["iters"], np.array([r["iters"] for r in ref]))
    assert np.array_equal(out["bits"], np.stack([r["x"] for r in ref]))
    assert np.array_equal(out["synd"], np.stack([r["z"] for r in ref]))
    assert any(r["rc"] == 0 for r in ref) and any(r["rc"] != 0 for r in ref)


@pytest.mark.parametrize("fname,q,F,perr", [("peg_4376_4_0.9_NB.txt", 13, 9, 0.012),
                                            ("q4.sp.9000.6000.4500.1", 4, 16, 0.12)])
def test_gfq_direct_probability_domain_equals_literal(hip, fname, q, F, perr):
    """The default check-node kernel carries sigma/rho as probabilities (one exp per edge
    symbol); cn_variant=1 evaluates the reference's log-sum-exp (GFq_LDPC.c:377-382) term by
    term.  Same decoded symbols, iteration counts and verdicts."""
    code = hip.Code.from_alist(code_path(fname))
    rng = np.random.default_rng(5)
    conf = _conf_matrix(q, perr)
    post = conf / conf.sum(axis=1, keepdims=True)                   # P(sent | rec), uniform prior
    x = rng.integers(0, q, size=(F, code.n))
    u = rng.random((F, code.n))
    pe = np.where(np.arange(F) % 3 == 2, 5 * perr, perr)[:, None]
    y = np.where(u < pe / 2, (x + 1) % q, np.where(u < pe, (x - 1) % q, x))
    logfna = np.log(np.maximum(post[y], 3.7e-44))                   # [F][n][q], floor exp(-100)
    s = np.stack([code.syndrome(x[f]) for f in range(F)]).astype(np.uint8)
    outs = []
    for variant in (0, 1):
        dec = hip.Decoder(code, hip.ALG.GFQ_DIRECT, max_frames=8, cn_variant=variant)
        outs.append(dec.decode(np.ascontiguousarray(logfna), s, 10))
    for k in ("status", "iters", "bits", "synd"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert (outs[0]["status"] == 0).any() and (outs[0]["status"] != 0).any()


def test_gfq_ldpc_harness(hip):
    """GFq_LDPC() (GFq_LDPC.c:540-655) through the reference-shaped entry point."""
    import ctypes as C
    L = hip.load_library()
    L.ldpc_set_seed(1234)
    q = 13
    conf = np.ascontiguousarray(_conf_matrix(q, 0.01) * 3.0)      # un-normalised on purpose
    err = (C.c_double * 2)()
    rc = L.GFq_LDPC(code_path("peg_4376_4_0.9_NB.txt").encode(), 15, 6, 0, 1, 0,
                    conf.ctypes.data_as(C.POINTER(C.c_double)), err)
    assert rc == 0, L.ldpc_last_error()
    assert err[0] == 0 and err[1] == 0
    assert np.allclose(conf.sum(axis=0), 1.0)                    # normalised in place, :548-558
    conf2 = np.ascontiguousarray(_conf_matrix(q, 0.5))
    rc = L.GFq_LDPC(code_path("peg_4376_4_0.9_NB.txt").encode(), 3, 4, 0, 1, 0,
                    conf2.ctypes.data_as(C.POINTER(C.c_double)), err)
    assert rc == 0 and err[0] == 4
    rc = L.GFq_LDPC(code_path("peg_4376_4_0.9_NB.txt").encode(), 3, 4, 0, 1, 1,
                    conf2.ctypes.data_as(C.POINTER(C.c_double)), err)
    assert rc == -6


@pytest.mark.parametrize("q,fname,channel,level", [(4, "q4.sp.9000.6000.4500.1", "bsc", 0.08),
                                                   (8, "q8.sp.6000.4000.3000.1", "lap", 0.5)])
def test_GFq_LDPC_NTT_program(hip, q, fname, channel, level):
    """GFq_LDPC_NTT(): the reference program's main() (:641-703) on its own usage examples
    (:660-661) with -iter 50; verdicts equal those of the compiled reference
    (tests/golden/ref_ntt_q*.npz rows 0..2: "converged." three times)."""
    g = np.load(os.path.join(GOLDEN, f"ref_ntt_q{q}.npz"))
    L = hip.load_library()
    verdict = np.full(3, 7, np.int32)
    iters = np.zeros(3, np.int32)
    rc = L.GFq_LDPC_NTT(int(g["maxit"]), channel.encode(), level, code_path(fname).encode(),
                        verdict.ctypes.data, iters.ctypes.data)
    assert rc == 0, L.ldpc_last_error()
    assert np.array_equal(verdict, g["rc"][:3])
    assert ((iters >= 1) & (iters <= int(g["maxit"]))).all()
    # one iteration is not enough for any of them: dec() returns -1 (:543)
    rc = L.GFq_LDPC_NTT(1, channel.encode(), level, code_path(fname).encode(), verdict.ctypes.data, None)
    assert rc == 0 and (verdict == -1).all()
    assert L.GFq_LDPC_NTT(5, b"awgn", level, code_path(fname).encode(), verdict.ctypes.data, None) < 0


def test_gfq_ldpc_graph_passing_boundary(hip):
    """GFq_LDPC() of WD_farm/GFq_LDPC.c:477: the graph handed over as the flat arrays of
    InitSumProductDecoderSyndrome.m instead of a file.  Same seed -> same trials -> the same
    [failures, undetected] as the file-name entry point; inconsistent arrays are refused."""
    import ctypes as C
    L = hip.load_library()
    path = code_path("peg_4376_4_0.9_NB.txt")
    code = hip.Code.from_alist(path)
    rp, ci, hv = code.csr()
    n, m, q = code.n, code.m, code.q
    rw = np.diff(rp).astype(np.int32)
    rmax = int(rw.max())
    rlc = np.zeros((m, rmax), np.int32)
    H = np.zeros((m, rmax), np.int32)
    cw = np.zeros(n, np.int32)
    cmax = int(np.bincount(ci, minlength=n).max())
    clr = np.zeros((n, cmax), np.int32)
    clN = np.zeros((n, cmax), np.int32)
    for j in range(m):
        for k in range(rw[j]):
            v = ci[rp[j] + k]
            rlc[j, k], H[j, k] = v, hv[rp[j] + k]
            clr[v, cw[v]], clN[v, cw[v]] = j, k
            cw[v] += 1
    ip = C.POINTER(C.c_int)
    dp = C.POINTER(C.c_double)
    P = lambda a: a.ctypes.data_as(ip)
    res = []
    for p_sym, iters, trials in ((0.01, 15, 5), (0.5, 3, 3)):
        conf = np.ascontiguousarray(_conf_matrix(q, p_sym))
        e1, e2 = (C.c_double * 2)(), (C.c_double * 2)()
        L.ldpc_set_seed(99)
        assert L.GFq_LDPC(path.encode(), iters, trials, 0, 1, 0, conf.copy().ctypes.data_as(dp), e1) == 0
        L.ldpc_set_seed(99)
        c2 = conf.copy()
        rc = L.GFq_LDPC_graph(q, iters, trials, 0, 1, 0, c2.ctypes.data_as(dp), n, m, cmax, rmax, P(cw), P(rw), P(rlc),
                              P(clr), P(clN), P(H), e2)
        assert rc == 0, L.ldpc_last_error()
        assert list(e1) == list(e2)
        res.append(list(e2))
    assert res[0] == [0.0, 0.0] and res[1][0] == 3.0
    bad = clN.copy()
    bad[7, 0] += 1
    e = (C.c_double * 2)()
    conf = np.ascontiguousarray(_conf_matrix(q, 0.01))
    assert L.GFq_LDPC_graph(q, 3, 1, 0, 1, 0, conf.ctypes.data_as(dp), n, m, cmax, rmax, P(cw), P(rw), P(rlc), P(clr),
                            P(bad), P(H), e) < 0
    assert b"col_list" in L.ldpc_last_error()


@pytest.mark.parametrize("fname,q,num_reads,alg", [("q4.sp.9000.6000.4500.1", 4, 1, "GFQ_DIRECT"),
                                                    ("peg_4376_4_0.9_NB.txt", 13, 3, "GFQ_DIRECT"),
                                                    ("peg_4376_4_0.9_NB.txt", 13, 1, "GFQ_DIRECT")])
def test_gfq_trial_loop_on_the_device(hip, oracle, fname, q, num_reads, alg):
    """The GF(q) Monte-Carlo front-end (ldpc_gfmc_*, mc_frontend.hip): the frames the device generates --
    source symbols, reads, the uniform draws behind them, log-probabilities, syndromes -- equal the host
    restatement GFq_LDPC_experiment() bit for bit (same counter-based draws), and that restatement is the
    one tests/test_host_library.py holds against the oracle's channel(), draw by draw; the syndromes
    also against the oracle's enc() here.  Then the counters of a multi-pass run equal what decoding the
    same frames through ldpc_decode_batch and classifying on the host gives, and frames do not depend on
    the pass size (first_frame)."""
    import ctypes as C
    L = hip.load_library()
    rng = np.random.default_rng(q + num_reads)
    R = q * num_reads
    conf = rng.random((R, q)) * (0.0015 if q == 13 else 0.08)
    for a in range(q):
        conf[a * num_reads:(a + 1) * num_reads, a] += 1.0 / num_reads   # reads of the sent symbol dominate
    conf[rng.random((R, q)) < 0.2] = 0.0
    for a in range(q):
        conf[a * num_reads, a] = max(conf[a * num_reads, a], 0.5)
    norm = conf / conf.sum(axis=0, keepdims=True)
    code = hip.Code.from_alist(code_path(fname))
    n, m = code.n, code.m
    dec = hip.Decoder(code, getattr(hip.ALG, alg), max_frames=24, early_exit=True)
    mc = hip.GfMonteCarlo(code, dec)
    mc.set_channel(norm, num_reads)
    F, seed = 5, 424242
    x, y, sel, lf, s = mc.generate(F, seed=seed, first_frame=3)
    # the host restatement of the same frames (frames 0..7 of the seed; the device was asked for 3..7)
    hx = np.zeros((8, n), np.int32); hs = np.zeros((8, m), np.int32); hy = np.zeros((8, n), np.int32)
    hl = np.zeros((8, n, q), np.float64); hsel = np.zeros((8, n), np.float32)
    mine = np.ascontiguousarray(conf, np.float64).copy()
    rc = L.GFq_LDPC_experiment(code_path(fname).encode(), num_reads, mine.ctypes.data_as(C.POINTER(C.c_double)), seed, 8,
                               hx.ctypes.data, hs.ctypes.data, hy.ctypes.data, hl.ctypes.data, hsel.ctypes.data)
    assert rc == 0, L.ldpc_last_error()
    assert np.allclose(mine, norm, rtol=0, atol=1e-15)
    assert np.array_equal(x, hx[3:].astype(np.uint8)) and np.array_equal(y, hy[3:]) and np.array_equal(sel, hsel[3:])
    assert np.array_equal(s, hs[3:].astype(np.uint8))
    assert np.array_equal(lf, hl[3:]), "log-probabilities differ from the host restatement"
    oc = oracle.load(code_path(fname))
    for f in range(F):
        assert np.array_equal(s[f], oracle.gf_enc(oc, q, x[f].astype(np.int32)))
    assert (lf == -100).any() and len(np.unique(y)) > q // 2
    # counters of a run in passes of 24 frames against the host-side classification of the same frames
    total, it = 60, 12
    c = mc.run(total, it, seed=seed)
    hx = np.zeros((total, n), np.int32); hl = np.zeros((total, n, q), np.float64); hs = np.zeros((total, m), np.int32)
    mine = np.ascontiguousarray(conf, np.float64).copy()
    rc = L.GFq_LDPC_experiment(code_path(fname).encode(), num_reads, mine.ctypes.data_as(C.POINTER(C.c_double)), seed, total,
                               hx.ctypes.data, hs.ctypes.data, None, hl.ctypes.data, None)
    assert rc == 0
    out = dec.decode(hl.reshape(total, n * q), hs.astype(np.uint8), it)
    nconv = int((out["status"] != 0).sum())
    print(f"gfmc {fname} reads={num_reads}: {nconv} of {total} frames fail, iterations {sorted(set(out['iters'].tolist()))}")
    wrong = (out["bits"] != hx.astype(np.uint8)).any(axis=1)
    assert c["frames"] == total and c["not_converged"] == nconv
    assert c["undetected"] == int(((out["status"] == 0) & wrong).sum())
    assert c["symbol_errors"] == int((out["bits"] != hx.astype(np.uint8)).sum())
    assert c["iterations"] == int(out["iters"].sum())
    # the last pass (frames 48..59) is still on the device
    lx, lxh, lst, lit = mc.last_pass(12)
    assert np.array_equal(lx, hx[48:].astype(np.uint8)) and np.array_equal(lxh, out["bits"][48:])
    assert np.array_equal(lst, out["status"][48:]) and np.array_equal(lit, out["iters"][48:])
    # sharding: the second half of the run as its own call
    c2 = mc.run(total - 30, it, seed=seed, first_frame=30)
    assert c2["not_converged"] == int((out["status"][30:] != 0).sum())
    assert c2["symbol_errors"] == int((out["bits"][30:] != hx[30:].astype(np.uint8)).sum())
    mc.close()
    dec.close()


@pytest.mark.parametrize("alg", ["GFQ_DIRECT", "GF2P_WHT"])
def test_gf_two_stage_decode_equals_one_stage(hip, monkeypatch, alg):
    """Early exit that frees work (GfEngine::decode, round 3): with most frames done after a few iterations and a
    few that need many (or fail), the first stage stops at 1/16 open frames and the rest are decoded again as a
    dense batch.  dec() is deterministic (GFq_LDPC.c:356-467, GFq_LDPC_NTT.c:437-544), so words, syndromes,
    iteration counts and verdicts must equal the one-stage decode's, frame for frame."""
    fname, q = "q4.sp.9000.6000.4500.1", 4
    code = hip.Code.from_alist(code_path(fname))
    n, m = code.n, code.m
    F = 512
    rng = np.random.default_rng(99)
    x = rng.integers(0, q, size=(F, n)).astype(np.int32)
    s = np.stack([code.syndrome(x[f]) for f in range(F)]).astype(np.uint8)
    # symbol errors: a light channel for most frames, a heavy one for every 23rd (slow or failing frames)
    perr = np.where(np.arange(F) % 23 == 0, np.where(np.arange(F) % 46 == 0, 0.26, 0.215), 0.06)
    err = rng.random((F, n)) < perr[:, None]
    y = np.where(err, (x + rng.integers(1, q, size=(F, n))) % q, x)
    pr = np.full((F, n, q), 0.0)
    for a in range(q):
        pr[:, :, a] = np.where(y == a, 1 - 0.09, 0.09 / (q - 1))
    if alg == "GF2P_WHT":
        llr = np.round(np.log2(pr) * 16384).astype(np.int32)
    else:
        llr = np.log(pr)
    dec = hip.Decoder(code, getattr(hip.ALG, alg), max_frames=F, early_exit=True)
    it = 40
    two = dec.decode(llr.reshape(F, n * q), s, it)
    launches_two = dec.last_iterations
    monkeypatch.setenv("LDPC_HIP_TWO_STAGE", "0")
    one = dec.decode(llr.reshape(F, n * q), s, it)
    monkeypatch.delenv("LDPC_HIP_TWO_STAGE")
    slow = (one["iters"] > 16) | (one["status"] != 0)
    print(f"{alg}: {int((one['status'] != 0).sum())} frames fail, {int(slow.sum())} slow, iterations "
          f"{int(np.median(one['iters']))} median / {int(one['iters'].max())} max; passes: {launches_two}")
    assert 0 < slow.sum() <= F // 16, "the case should leave a few (at most 1/16) slow frames"
    for k in ("bits", "synd", "iters", "status"):
        assert np.array_equal(one[k], two[k]), k
    dec.close()
