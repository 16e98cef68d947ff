"""CPU: the C-ABI library loads, exports every symbol include/ldpc_hip.h declares, and
its host logic (alist loader, CSR/CSC, syndrome, error codes) agrees with the oracle's
restatement of the reference loader.  No compute calls: there is no GPU here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import CODES, ROOT, code_path

ALL_CODES = sorted(f for f in os.listdir(CODES))


def test_exports_every_declared_symbol(hip):
    L = hip.load_library()
    hdr = open(os.path.join(ROOT, "include", "ldpc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", hdr))
    names = {n for n in names if n.startswith(("ldpc_", "test_code", "GFq_LDPC"))}
    assert len(names) >= 30
    for n in sorted(names):
        assert hasattr(L, n), f"libldpc_hip.so does not export {n}"
    assert b"gfx950" in L.ldpc_version()


@pytest.mark.parametrize("name", ALL_CODES)
def test_loader_matches_reference_edge_numbering(hip, oracle, name):
    """Edges keep the reference numbering: CSR = row lists in file order, column slot k
    of variable v = k-th row (ascending) holding v (test_code_min_sum_B.c:253-264)."""
    code = hip.Code.from_alist(code_path(name))
    if name.startswith("dvb"):
        # the reference's fscanf loaders misparse this unpadded file (hard part 5);
        # check internal consistency only
        rp, ci, hv = code.csr()
        assert code.n == 64800 and code.m == 32400 and code.E == 226799
        assert (np.diff(rp) >= 6).all() and np.diff(rp).max() == 7
    else:
        # cmax-padded column lists for every shipped file except the col_weight dialect
        oc = None
        for dialect in (0, 1):
            try:
                oc = oracle.load(code_path(name), dialect)
                break
            except OSError:
                continue
        assert oc is not None
        assert (code.n, code.m, code.q) == (oc.n, oc.m, oc.q)
        rp, ci, hv = code.csr()
        cp, eo = code.csc()
        assert code.cmax == oc.col_weight.max() and code.rmax == oc.row_weight.max()
        assert np.array_equal(np.diff(rp), oc.row_weight)
        assert np.array_equal(np.diff(cp), oc.col_weight)
        for j in range(oc.m):
            w = oc.row_weight[j]
            assert np.array_equal(ci[rp[j]:rp[j] + w], oc.row_col[j, :w])
            assert np.array_equal(hv[rp[j]:rp[j] + w], oc.H[j, :w])
        for v in range(0, oc.n, max(1, oc.n // 997)):
            for k in range(oc.col_weight[v]):
                e = eo[cp[v] + k]
                j = oc.col_row[v, k]
                assert rp[j] + oc.col_N[v, k] == e
    # syndrome of a random word agrees with the oracle's enc()
    if not name.startswith("dvb"):
        rng = np.random.default_rng(1)
        x = rng.integers(0, code.q, code.n).astype(np.int32)
        want = oracle.enc(oc, x) if code.q == 2 else oracle.gf_enc(oc, code.q, x)
        assert np.array_equal(code.syndrome(x), want)


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


SMALL = """4 2
2 3
1 1 2 1
3 2
1 0
1 0
1 2
2 0
1 2 3
3 4 0
"""


def test_alist_dialects(hip, tmp_path):
    a = hip.Code.from_alist(_write(tmp_path, "padded.txt", SMALL))
    rp, ci, _ = a.csr()
    assert rp.tolist() == [0, 3, 5] and ci.tolist() == [0, 1, 2, 2, 3]
    assert a.csc()[1].tolist() == [0, 1, 2, 3, 4] and a.csc()[0].tolist() == [0, 1, 2, 4, 5]
    # unpadded rows and columns, tabs, CRLF, trailing blanks
    un = SMALL.replace("3 4 0", "3\t4  ").replace("1 0\n1 0\n1 2\n2 0\n", "1\n1\n1 2\n2\n")
    b = hip.Code.from_alist(_write(tmp_path, "unpadded.txt", un.replace("\n", "\r\n")))
    assert np.array_equal(b.csr()[1], ci)
    # everything as one token stream (the reference's fscanf view): padded, and unpadded
    c = hip.Code.from_alist(_write(tmp_path, "stream.txt", " ".join(SMALL.split())))
    assert np.array_equal(c.csr()[1], ci)
    d = hip.Code.from_alist(_write(tmp_path, "stream2.txt", " ".join(un.split())))
    assert np.array_equal(d.csr()[1], ci) and d.cmax == 2 and d.rmax == 3


def test_alist_errors(hip, tmp_path):
    from ldpc_amd import LdpcError
    with pytest.raises(LdpcError) as e:
        hip.Code.from_alist(str(tmp_path / "missing.txt"))
    assert e.value.code == -2                       # reference: exit(-2)
    with pytest.raises(LdpcError) as e:
        hip.Code.from_alist(_write(tmp_path, "bad.txt", SMALL.replace("3 4 0", "3 9 0")))
    assert e.value.code == -1
    with pytest.raises(LdpcError) as e:
        hip.Code.from_alist(_write(tmp_path, "short.txt", SMALL[:30]))
    assert e.value.code == -1
    nb = "3 1 4\n1 3\n1 1 1\n3\n1 2\n1 3\n1 1\n1 2 2 3 3 1\n"
    g = hip.Code.from_alist(_write(tmp_path, "nb.txt", nb))
    assert g.q == 4 and g.csr()[2].tolist() == [2, 3, 1]
    with pytest.raises(LdpcError):                  # non-zero filler, GFq_LDPC.c:517-524
        hip.Code.from_alist(_write(tmp_path, "nbbad.txt",
                                   "3 1 4\n1 3\n1 1 0\n2\n1 2\n1 3\n0 0\n1 2 2 3 7 7\n".replace("1 1 0\n2", "1 1 0\n2")))


def test_matrix_alist_roundtrip(hip, tmp_path):
    from ldpc_amd.alist import alist_to_mat, matrix_to_alist
    rng = np.random.default_rng(3)
    H = (rng.random((7, 15)) < 0.3).astype(int)
    H[:, H.sum(0) == 0] = np.eye(7, dtype=int)[:, :1]
    H[H.sum(1) == 0, 0] = 1
    p = str(tmp_path / "h.alist")
    matrix_to_alist(H, p)
    assert np.array_equal(alist_to_mat(p), H)
    code = hip.Code.from_dense(H)
    code2 = hip.Code.from_alist(p)
    assert all(np.array_equal(a, b) for a, b in zip(code.csr(), code2.csr()))
    assert all(np.array_equal(a, b) for a, b in zip(code.csc(), code2.csc()))


def test_no_gpu_is_a_loud_error(hip):
    """Without a HIP device a decoder cannot be created: no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ldpc_amd import LdpcError
    code = hip.Code.from_alist(code_path("small_code.txt"))
    with pytest.raises(LdpcError) as e:
        hip.Decoder(code, hip.ALG.MINSUM_Q88)
    assert e.value.code == -5
    L = hip.load_library()
    err = (C.c_double * 2)()
    rc = L.test_code_min_sum_B_cw_noise_gen(code_path("small_code.txt").encode(), 5, 3, 0.01, err)
    assert rc == -5
    rc = L.test_code_B(5, 3, 0.01, b"/nonexistent.alist", err)
    assert rc == -2


def test_product_does_not_import_oracle():
    """The product path must never route through oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ldpc_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), f"{f} mentions the oracle"


@pytest.mark.parametrize("name,block,free", [("peg_16000_3_0.9.txt", 1024, True), ("204.33.484.txt", 128, False),
                                             ("Frolov_1024_0.5.txt", 512, True), ("4095.737.3.txt", 768, True),
                                             ("32000.2240.3.105.txt", 1024, True)])
def test_lds_schedule_is_a_permutation_and_conflict_free(hip, name, block, free):
    """The fused decoder's host-side placement (lds_schedule.cpp): every edge of every check
    appears in exactly one slot (errors == 0: the self check inside the library rebuilds the
    edge multiset from the tables) and, with optimisation on, no 32-lane group ever meets a
    bank twice (`free`; a check-regular code whose slot count has no slack, like (204,102)
    with 6 slots for weight-6 checks, leaves the colouring no room and keeps a few); the
    natural placement is the comparison."""
    code = hip.Code.from_alist(code_path(name))
    opt = code.lds_schedule_stats(block, True)
    nat = code.lds_schedule_stats(block, False)
    assert opt["errors"] == 0 and nat["errors"] == 0
    assert opt["accesses"] == nat["accesses"] > 0
    assert nat["conflicts"] >= opt["conflicts"]      # (the quasi-cyclic Frolov code is conflict free as it stands)
    if free:
        assert opt["conflicts"] == 0
    assert opt["words"] <= 65535


@pytest.mark.parametrize("name,block", [("peg_16000_3_0.9.txt", 1024), ("4095.737.3.txt", 512),
                                        ("1057.244.3.txt", 256), ("Frolov_1024_0.5.txt", 512)])
def test_variable_pass_placement_keeps_the_schedule_valid(hip, name, block):
    """vpass_place.cpp re-ranks variables within their LDS banks for the fused floating-point
    min-sum kernel's variable pass.  It must leave a valid schedule (errors == 0: every variable
    still alone on a word of its own bank, every check still visiting exactly its variables,
    pad slots untouched -- so the check pass stays conflict free) and must not make the
    modelled bank-conflict cycles of the variable pass worse; on the regular codes it removes
    more than half of them."""
    code = hip.Code.from_alist(code_path(name))
    st = code.vpass_place_stats(block)
    assert st["errors"] == 0
    assert st["reads"] > 0 and 0 <= st["after"] <= st["before"]
    if not name.startswith("Frolov"):
        assert st["after"] < 0.5 * st["before"]


@pytest.mark.parametrize("q,fname,channel,level", [(4, "q4.sp.9000.6000.4500.1", "bsc", 0.08),
                                                   (8, "q8.sp.6000.4000.3000.1", "lap", 0.5)])
def test_ntt_program_experiments_match_reference(hip, q, fname, channel, level):
    """GFq_LDPC_NTT_experiment (host only) regenerates the inputs of the reference program's
    three experiments (main(), GFq_LDPC_NTT.c:681-687: LCG source word, enc(), bsc()/lap())
    exactly: compared with x, s, logfna captured from the compiled reference
    (tests/golden/ref_ntt_q*.npz, rows 0..2 = seeds 1..3)."""
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, f"ref_ntt_q{q}.npz"))
    L = hip.load_library()
    n, m = g["x"].shape[1], g["s"].shape[1]
    for seed in (1, 2, 3):
        x = np.zeros(n, np.int32); s = np.zeros(m, np.int32); y = np.zeros(n, np.int32)
        lf = np.zeros((n, q), np.int32)
        rc = L.GFq_LDPC_NTT_experiment(code_path(fname).encode(), seed, channel.encode(), level,
                                       x.ctypes.data, s.ctypes.data, y.ctypes.data, lf.ctypes.data)
        assert rc == 0, L.ldpc_last_error()
        assert np.array_equal(x, g["x"][seed - 1])
        assert np.array_equal(s, g["s"][seed - 1])
        assert np.array_equal(lf, g["logfna"][seed - 1])
        assert (y != x).any() and ((y >= 0) & (y < q)).all()
    # argument errors come back as codes, not exit()/usage text
    assert L.GFq_LDPC_NTT_experiment(code_path(fname).encode(), 1, b"awgn", level, x.ctypes.data, s.ctypes.data,
                                     y.ctypes.data, lf.ctypes.data) < 0
    assert L.GFq_LDPC_NTT_experiment(code_path("peg_4376_4_0.9_NB.txt").encode(), 1, b"bsc", level, x.ctypes.data,
                                     s.ctypes.data, y.ctypes.data, lf.ctypes.data) < 0     # GF(13): not 2^p


def test_c_drivers_fail_loudly_without_a_gpu():
    """The plain-C driver programs (gcc, no HIP headers) link against the library; where no
    HIP device exists they exit non-zero with the library's message -- no CPU decode path."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by tests/test_gpu_entry_points.py")
    exe = os.path.join(ROOT, "tools", "drivers", "bin", "test_code_no_mex_B")
    assert os.path.exists(exe), "run __graft_entry__.build() (make -C tools/drivers)"
    r = subprocess.run([exe, code_path("4095.737.3.txt")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "no HIP device" in r.stderr and r.stdout == ""
    exe = os.path.join(ROOT, "tools", "drivers", "bin", "GFq_LDPC_NTT")
    r = subprocess.run([exe, "bsc", "0.08", code_path("q4.sp.9000.6000.4500.1")], capture_output=True, text=True,
                       timeout=60)
    assert r.returncode != 0 and "no HIP device" in r.stderr and r.stdout == ""
    r = subprocess.run([exe, "bsc"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and r.stderr.startswith("usage:")


@pytest.mark.parametrize("fname,q,num_reads", [("peg_4376_4_0.9_NB.txt", 13, 1), ("peg_4376_4_0.9_NB.txt", 13, 2),
                                               ("q4.sp.9000.6000.4500.1", 4, 3)])
def test_gfq_channel_front_end_equals_the_restated_reference(hip, oracle, fname, q, num_reads):
    """GFq_LDPC()'s channel front-end (GFq_LDPC.c:177-248: column normalisation :548-558,
    make_p_sent_given_rec_T, channel()'s walk down the column of the sent symbol, assign_llr
    with the -100 floor) as the library prepares it on the host, against the oracle's
    restatement, bit for bit: same reads y from the same uniform draws, same log-probabilities
    (glibc log on both sides).  The reference draws from rand() seeded by the clock, so the
    draws themselves are an input here (GFq_LDPC_experiment hands them out).  Host only."""
    import ctypes as C
    L = hip.load_library()
    rng = np.random.default_rng(q * 10 + num_reads)
    R = q * num_reads
    conf = rng.random((R, q)) ** 4                          # skewed, un-normalised
    conf[rng.random((R, q)) < 0.3] = 0.0                    # zeros: the -100 floor and the skipped bins
    conf[np.arange(q) % R, np.arange(q)] += 2.0             # every column keeps mass
    if num_reads > 1:
        conf[R - 1] = 0.0                                   # a read value that never occurs (P_y = 0 branch)
    mine = np.ascontiguousarray(conf, np.float64).copy()
    code = hip.Code.from_alist(code_path(fname))
    n, m, F = code.n, code.m, 3
    x = np.zeros((F, n), np.int32); s = np.zeros((F, m), np.int32); y = np.zeros((F, n), np.int32)
    lf = np.zeros((F, n, q), np.float64); sel = np.zeros((F, n), np.float32)
    rc = L.GFq_LDPC_experiment(code_path(fname).encode(), num_reads, mine.ctypes.data_as(C.POINTER(C.c_double)),
                               987654321, F, x.ctypes.data, s.ctypes.data, y.ctypes.data, lf.ctypes.data,
                               sel.ctypes.data)
    assert rc == 0, L.ldpc_last_error()
    theirs = np.ascontiguousarray(conf, np.float64).copy()
    oracle.lib.ora_gf_normalise_confusion(q, num_reads, theirs)
    assert np.array_equal(mine, theirs)                     # normalised in place, :548-558
    oc = oracle.load(code_path(fname))
    assert ((x >= 0) & (x < q)).all() and len(np.unique(x)) == q
    assert (sel >= 0).all() and (sel <= 100).all() and sel.std() > 20
    for f in range(F):
        assert np.array_equal(s[f], oracle.gf_enc(oc, q, x[f]))
        yo = oracle.gf_channel_draw(q, num_reads, theirs, x[f], sel[f])
        assert np.array_equal(y[f], yo)
        lo = oracle.gf_assign_llr(q, num_reads, conf, yo)    # [q][n]; normalises its own copy
        assert np.array_equal(lf[f], lo.T)
    assert (lf == -100).any() and (y != x).any()
    if num_reads > 1:
        assert (y < R - 1).all()                            # the impossible read is never drawn


def test_malformed_csr_is_refused(hip):
    """ldpc_code_from_csr (the WD-farm graph-passing boundary, WD_farm/GFq_LDPC.c:477) validates
    the caller's arrays before anything is read through them: row_ptr must start at 0 and be
    monotone, columns in range, no variable twice in a row (the slot tables match edges by
    position: a duplicate would silently decode another graph)."""
    ok = dict(n=6, m=3, row_ptr=[0, 3, 5, 8], col_idx=[0, 1, 2, 2, 3, 3, 4, 5])
    code = hip.Code.from_csr(**ok)
    assert (code.n, code.m, code.E, code.rmax, code.cmax) == (6, 3, 8, 3, 2)
    bad = [dict(ok, row_ptr=[1, 3, 5, 8]),                      # leading edges that belong to no check
           dict(ok, row_ptr=[0, 5, 3, 8]),                      # not monotone
           dict(ok, col_idx=[0, 1, 1, 2, 3, 3, 4, 5]),          # variable 1 twice in row 0
           dict(ok, col_idx=[0, 1, 2, 2, 3, 3, 4, 6]),          # column out of range
           dict(ok, col_idx=[0, 1, 2, 2, 3, 3, 4, -1])]
    for kw in bad:
        with pytest.raises(hip.LdpcError) as e:
            hip.Code.from_csr(**kw)
        assert e.value.code == -1, kw
    with pytest.raises(hip.LdpcError):                          # GF coefficient out of range
        hip.Code.from_csr(4, 2, [0, 2, 4], [0, 1, 2, 3], hval=[1, 2, 3, 4], q=4)


def test_compact_wht_tables_are_exact(hip):
    """The GF(2^p) kernels evaluate Fexp / Flog (GFq_LDPC_NTT.c:196-247) from compact tables in LDS.
    Their arithmetic, restated on the host, must reproduce all 2^21 entries of fexp and all 2^20
    entries of flog of initlogexptab2() -- exactly, or by handing the argument back to the full
    table (a few hundred Flog arguments within 1e-4 of a rounding boundary)."""
    st = np.zeros(4, np.int64)
    bad = hip.load_library().ldpc_wht_tables_selfcheck(st.ctypes.data)
    assert bad == 0 and st[0] == 0 and st[1] == 0
    assert 0 < st[2] < 1000 and st[3] == 2048


def test_compact_spa_tables_are_exact(hip):
    """The table sum-product's check pass reads Flogtanh / Fgallag (inittab(), test_code_B.c:52-90) from a
    153 KiB LDS image (u32 below entry 2 048, u16 below 16 384, one bit per entry above).
    The kernels' arithmetic, restated on the host, must reproduce all 2^22 entries of both tables."""
    st = np.zeros(4, np.int64)
    bad = hip.load_library().ldpc_spa_tables_selfcheck(st.ctypes.data)
    assert st[0] == 1, "the image builder refused this libm's tables"
    assert bad == 0 and st[1] == 0 and st[2] == 0
    assert st[3] == 156480 <= 160 * 1024
