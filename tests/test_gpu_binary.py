"""GPU parity of the binary decoders against the CPU oracle (through the C-ABI).

Bit-exact for MINSUM_Q88 / MINSUM_F64 / MINSUM_F32 / SPA_TABLE_Q14 (decoded
bits, iteration counts, syndromes, status); decoded-bit parity with a margin
filter for the libm-dependent SPA_F64 / SPA_F32.
"""
import numpy as np
import pytest

from conftest import code_path, make_frames, to_q88

pytestmark = pytest.mark.gpu

CASES = [
    # code, F, p (bsc), sigma (awgn), max_iter
    ("small_code.txt", 40, 0.08, 0.9, 8),
    ("204.33.484.txt", 300, 0.04, 0.75, 20),
    ("Frolov_1024_0.5.txt", 70, 0.045, 0.8, 30),
    ("1057.244.3.txt", 530, 0.012, 0.55, 15),
    ("4095.737.3.txt", 33, 0.008, 0.5, 12),
    ("peg_16000_3_0.9.txt", 12, 0.0012, 0.42, 10),
    ("32000.2240.3.105.txt", 6, 0.004, 0.5, 8),
]


def run_oracle(fn, ocode, llrs, s, it):
    res = [fn(ocode, llrs[f], s[f], it) for f in range(len(llrs))]
    return (np.stack([r["bits"] for r in res]).astype(np.uint8),
            np.array([r["iters"] for r in res], np.int32),
            np.stack([r["synd"] for r in res]).astype(np.uint8),
            np.array([r["rc"] for r in res], np.int32),
            np.stack([r["post"] for r in res]))


def check_exact(out, ref, tag):
    bits, iters, synd, status, _ = ref
    assert np.array_equal(out["status"], status), f"{tag}: status differs"
    assert np.array_equal(out["iters"], iters), f"{tag}: iteration counts differ"
    assert np.array_equal(out["bits"], bits), f"{tag}: decoded bits differ"
    assert np.array_equal(out["synd"], synd), f"{tag}: syndromes differ"


@pytest.mark.parametrize("name,F,p,sigma,it", CASES)
@pytest.mark.parametrize("kind", ["bsc", "awgn"])
def test_minsum_q88(hip, oracle, name, F, p, sigma, it, kind):
    oc = oracle.load(code_path(name))
    x, s, llr = make_frames(oracle, oc, F, p, 11, kind, sigma=sigma)
    q, raw = to_q88(llr)
    ref = run_oracle(lambda c, l, sy, k: oracle.dec_minsum_fixed(c, l, sy, k, None, oc.m + 1), oc, q, s, it)
    code = hip.Code.from_alist(code_path(name))
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=512, early_exit=True, want_llr=True)
    for inp in (q, raw, llr.astype(np.float64)):      # grid doubles, raw int16, unquantised doubles
        out = dec.decode(inp, s.astype(np.uint8), it, want=("bits", "iters", "synd", "status", "llr"))
        check_exact(out, ref, f"q88 {name} {kind} {inp.dtype}")
    # posterior of running frames = float_to_fix(sum)
    nc = ref[3] != 0
    if nc.any():
        assert np.array_equal(out["llr"][nc].astype(np.float64), ref[4][nc])
    # early_exit off must give the same answers
    dec2 = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=512, early_exit=False)
    check_exact(dec2.decode(raw, s.astype(np.uint8), it), ref, f"q88-noexit {name} {kind}")


@pytest.mark.parametrize("name,F,p,sigma,it", CASES)
@pytest.mark.parametrize("kind", ["bsc", "awgn"])
def test_minsum_f64(hip, oracle, name, F, p, sigma, it, kind):
    oc = oracle.load(code_path(name))
    x, s, llr = make_frames(oracle, oc, F, p, 12, kind, sigma=sigma)
    ref = run_oracle(oracle.dec_minsum, oc, llr, s, it)
    code = hip.Code.from_alist(code_path(name))
    dec = hip.Decoder(code, hip.ALG.MINSUM_F64, max_frames=256, want_llr=True)
    out = dec.decode(llr, s.astype(np.uint8), it, want=("bits", "iters", "synd", "status", "llr"))
    check_exact(out, ref, f"ms64 {name} {kind}")


@pytest.mark.parametrize("name,F,p,sigma,it", CASES)
@pytest.mark.parametrize("kind", ["bsc", "awgn"])
def test_minsum_f32(hip, oracle, name, F, p, sigma, it, kind):
    oc = oracle.load(code_path(name))
    x, s, llr = make_frames(oracle, oc, F, p, 13, kind, sigma=sigma)
    llr32 = llr.astype(np.float32)
    ref = run_oracle(oracle.dec_minsum_f32, oc, llr32, s, it)
    code = hip.Code.from_alist(code_path(name))
    dec = hip.Decoder(code, hip.ALG.MINSUM_F32, max_frames=512, want_llr=True)
    out = dec.decode(llr32, s.astype(np.uint8), it, want=("bits", "iters", "synd", "status", "llr"))
    check_exact(out, ref, f"ms32 {name} {kind}")
    nc = ref[3] != 0
    if nc.any():   # binary32 posterior sums are bit-identical too
        assert np.array_equal(out["llr"][nc], ref[4][nc].astype(np.float32))
    # against the reference's own double arithmetic: decoded bits agree wherever the
    # double posterior is not within rounding distance of zero
    # (BSC inputs are skipped: with all |LLR| equal, posteriors that cancel to exactly 0
    # in binary64 land on +-1 ulp in binary32, a tie the two precisions break differently)
    if kind == "awgn":
        ref64 = run_oracle(oracle.dec_minsum, oc, llr32.astype(np.float64), s, it)
        # verdict and iteration count on every frame; the decoded word where the reference converged (a frame
        # that fails in both precisions wanders: its bits after max_iter iterations are no point of comparison)
        same = (ref64[3] == out["status"]) & (ref64[1] == out["iters"]) & \
               ((ref64[3] != 0) | (ref64[0] == out["bits"]).all(axis=1))
        print(f"ms32 vs the double reference, {name}: {same.sum()}/{len(same)} frames equal (verdict, iterations, converged words)")
        assert same.all(), f"frames {np.where(~same)[0]} differ from the double reference"   # (measured: all, on every case)


# (small_code, n = 12, has weight-2 checks that pass +-L straight through: with equal-magnitude
# inputs variable sums cancel exactly and the reference's log tanh(0) hazard fires on half the
# frames at the p of the min-sum cases; a lower p keeps most frames clear of it)
SPA_CASES = [("small_code.txt", 40, 0.03, 0.9, 8)] + \
            [c for c in CASES[1:] if c[0] not in ("peg_16000_3_0.9.txt", "32000.2240.3.105.txt")] + \
            [("peg_16000_3_0.9.txt", 6, 0.003, 0.4, 10)]


@pytest.mark.parametrize("name,F,p,sigma,it", SPA_CASES)
@pytest.mark.parametrize("kind", ["bsc", "awgn"])
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_spa_float(hip, oracle, name, F, p, sigma, it, kind, prec):
    """The sum-product decoders against their oracle (test_code_B_float.c:134-199): on EVERY frame whose
    oracle posteriors hold no NaN, verdict, iteration count and every decoded bit must be EQUAL -- no
    allowance, and no margin filter on this check (round 2 measured the device's own exp / log routines
    against glibc on these cases: all 1 958 frames agree, profiles/r02_spa_agreement.txt; the test now
    demands what was measured).  Frames with the reference's NaN (its log tanh(0) = -inf hazard, :149,:168:
    the outcome depends on whether a sum cancels exactly) are counted and printed, and at least 95 % of
    them must agree as well.  The margin (frames whose oracle posteriors all keep |sum| > MARGIN) only
    selects the frames on which posterior LLR values are compared, to the relative tolerance RTOL."""
    MARGIN = {"f64": 1e-6, "f32": 1e-2}[prec]
    RTOL = {"f64": 1e-6, "f32": 2e-2}[prec]
    oc = oracle.load(code_path(name))
    x, s, llr = make_frames(oracle, oc, F, p * 1.6, 14, kind, sigma=sigma * 1.08)
    if prec == "f32":
        llr = llr.astype(np.float32)
        ref = run_oracle(oracle.dec_spa_f32, oc, llr, s, it)
        alg = hip.ALG.SPA_F32
    else:
        ref = run_oracle(oracle.dec_spa, oc, llr, s, it)
        alg = hip.ALG.SPA_F64
    code = hip.Code.from_alist(code_path(name))
    dec = hip.Decoder(code, alg, max_frames=256, want_llr=True)
    out = dec.decode(llr, s.astype(np.uint8), it, want=("bits", "iters", "synd", "status", "llr"))
    post = ref[4]
    nan = np.isnan(post).any(axis=1)
    safe = ~nan & (np.abs(np.where(np.isnan(post), 1.0, post)).min(axis=1) > MARGIN)
    agree = (out["status"] == ref[3]) & (out["iters"] == ref[1]) & (out["bits"] == ref[0]).all(axis=1)
    print(f"spa {prec} {name} {kind}: {agree[~nan].sum()}/{(~nan).sum()} NaN-free frames equal, {agree[nan].sum()}/{nan.sum()} "
          f"of the frames with the reference's NaN")
    assert (~nan).sum() >= 0.6 * F, f"{nan.sum()} of {F} frames hit the reference's NaN: the case tests too little"
    assert agree[~nan].all(), f"spa {prec} {name} {kind}: frames {np.where(~nan & ~agree)[0]} differ"
    assert agree[nan].sum() >= 0.95 * nan.sum() - 1e-9, f"spa {prec} {name} {kind}: NaN frames {np.where(nan & ~agree)[0]} differ"
    assert np.array_equal(out["synd"][~nan], ref[2][~nan])
    if prec == "f32":
        assert not np.isnan(out["llr"][~nan]).any(), "binary32 NaN on a frame where its twin has none"
    # posterior LLRs of frames that ran all iterations on both sides
    both = safe & (ref[3] != 0) & (out["status"] != 0) & agree
    if both.any():
        a, b = out["llr"][both].astype(np.float64), post[both].astype(np.float64)
        fin = np.isfinite(a) & np.isfinite(b)
        rel = np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), 1.0)
        assert np.median(rel) < RTOL


@pytest.mark.parametrize("name,F,p,sigma,it", CASES[:5])
def test_spa_table(hip, oracle, name, F, p, sigma, it):
    oc = oracle.load(code_path(name))
    x, s, llr = make_frames(oracle, oc, min(F, 64), p * 1.6, 15, "bsc")
    q14 = oracle.float2fix14(llr)
    ref = run_oracle(oracle.dec_spa_table, oc, q14, s, it)
    assert (ref[3] >= 0).all()
    code = hip.Code.from_alist(code_path(name))
    dec = hip.Decoder(code, hip.ALG.SPA_TABLE_Q14, max_frames=128)
    check_exact(dec.decode(q14, s.astype(np.uint8), it), ref, f"tab {name} i32")
    check_exact(dec.decode(llr, s.astype(np.uint8), it), ref, f"tab {name} f64")


def test_spa_table_on_ragged_rows(hip, oracle, tmp_path):
    """The table sum-product's LDS-table kernel takes a check's rows in chunks (one chunk when it holds the
    whole check, else two sweeps of double-buffered chunks whose last one is filled up with copies of the
    last row and whose last two are not read again): one code with every row weight from 1 to 45 walks
    through every remainder, for the three chunk sizes the kernel is built with."""
    import os
    rng = np.random.default_rng(4242)
    n = 360
    rows = [rng.choice(n, 1 + (j % 45), replace=False) for j in range(90)]
    rows += [np.arange(8 * j, 8 * j + 8) for j in range(n // 8)]          # every variable is in some check
    H = np.zeros((len(rows), n), np.uint8)
    for j, r in enumerate(rows):
        H[j, r] = 1
    path = str(tmp_path / "ragged.alist")
    hip.alist.matrix_to_alist(H, path)
    oc = oracle.load(path)
    code = hip.Code.from_alist(path)
    x, s, llr = make_frames(oracle, oc, 300, 0.03, 9, "bsc")
    q14 = oracle.float2fix14(llr)
    ref = run_oracle(oracle.dec_spa_table, oc, q14, s, 12)
    ok = ref[3] >= 0                  # (rc -2: the reference itself indexed outside its table)
    assert ok.sum() >= 250
    ref = tuple(r[ok] for r in ref)
    for ch in ("4", "8", "16"):
        os.environ["LDPC_HIP_TAB_CH"] = ch
        try:
            dec = hip.Decoder(code, hip.ALG.SPA_TABLE_Q14, max_frames=512, early_exit=True)
        finally:
            del os.environ["LDPC_HIP_TAB_CH"]
        out = dec.decode(q14, s.astype(np.uint8), 12)
        check_exact({k: v[ok] for k, v in out.items()}, ref, f"table ragged rows, chunks of {ch}")
    assert len(set(ref[1].tolist())) > 2, "frames should stop at different iterations"


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("name,F,p,it", [("204.33.484.txt", 200, 0.05, 15), ("peg_16000_3_0.9.txt", 9, 0.0015, 8),
                                         ("Frolov_1024_0.5.txt", 40, 0.045, 20)])
def test_every_check_node_kernel_is_bit_exact(hip, oracle, name, F, p, it, variant):
    """The two-pass engine's check-node kernels (opts.cn_variant 1 LDS staged, 2 reload,
    3 register resident, 4 DMA staged, 5 single read, 6 both tables in LDS (table sum-product only; min-sum
    then takes its default); 0 = the measured best per algorithm) all give the
    reference's results: Q8.8 (non-fused path), binary64 min-sum and the table sum-product."""
    oc = oracle.load(code_path(name))
    x, s, llr = make_frames(oracle, oc, F, p, 23, "bsc")
    q, raw = to_q88(llr)
    su = s.astype(np.uint8)
    code = hip.Code.from_alist(code_path(name))
    ref = run_oracle(lambda c, l, sy, k: oracle.dec_minsum_fixed(c, l, sy, k, None, oc.m + 1), oc, q, s, it)
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=256, early_exit=True, cn_variant=variant)
    check_exact(dec.decode(raw, su, it), ref, f"q88 cn_variant {variant} {name}")
    ref = run_oracle(oracle.dec_minsum, oc, llr, s, it)
    dec = hip.Decoder(code, hip.ALG.MINSUM_F64, max_frames=256, early_exit=False, cn_variant=variant)
    check_exact(dec.decode(llr.astype(np.float64), su, it), ref, f"f64 cn_variant {variant} {name}")
    if variant not in (3, 5):                         # (those two kernels are min-sum only)
        q14 = oracle.float2fix14(llr)
        ref = run_oracle(oracle.dec_spa_table, oc, q14, s, it)
        dec = hip.Decoder(code, hip.ALG.SPA_TABLE_Q14, max_frames=256, early_exit=True, cn_variant=variant)
        check_exact(dec.decode(q14, su, it), ref, f"table cn_variant {variant} {name}")


def _random_code(rng, n, m, dv):
    """Random H with column weight dv (rows filled as evenly as a random fill allows, no
    duplicate entries, every row weight >= 2)."""
    H = np.zeros((m, n), np.uint8)
    load = np.zeros(m, np.int64)
    for v in range(n):
        order = np.lexsort((rng.random(m), load))          # lightest rows first, random ties
        rows = order[:dv]
        H[rows, v] = 1
        load[rows] += 1
    assert (H.sum(axis=1) >= 2).all()
    return H


@pytest.mark.parametrize("n,m,dv,it", [(96, 48, 3, 10), (520, 130, 3, 12), (2000, 200, 3, 8), (3000, 150, 3, 8),
                                       (1200, 1150, 3, 15), (4000, 1000, 4, 8), (777, 259, 5, 10), (6000, 300, 2, 6),
                                       (2044, 1022, 6, 10), (6000, 4000, 3, 10), (5000, 3300, 2, 6),
                                       (12000, 2200, 4, 8)])
def test_fused_q88_on_random_shapes(hip, oracle, tmp_path, n, m, dv, it):
    """The fused decoder picks a kernel per (slots per check, checks per thread, address width):
    random codes of many shapes walk through those instantiations -- row weights 3 to 60, one
    to three checks per thread (12000 x 2200: the 12-wave build for three heavy checks per
    thread), n not a multiple of 8, and (m > 3072) the generic kernel with
    four and more checks per thread -- each bit-exact against the oracle, with non-zero
    syndromes and a mix of converging and failing frames; every input type is fed once."""
    rng = np.random.default_rng(n * 7 + m)
    H = _random_code(rng, n, m, dv)
    path = str(tmp_path / "rand.alist")
    hip.alist.matrix_to_alist(H, path)
    oc = oracle.load(path)
    code = hip.Code.from_alist(path)
    assert (code.n, code.m, code.E) == (n, m, n * dv)
    F = 24
    p = {2: 0.004, 3: 0.02, 4: 0.03, 5: 0.04, 6: 0.05}[dv] * (0.5 if n / m > 5 else 1.0)
    x, s, llr = make_frames(oracle, oc, F, p, 5, "bsc")
    q, raw = to_q88(llr)
    ref = run_oracle(lambda c, l, sy, k: oracle.dec_minsum_fixed(c, l, sy, k, None, oc.m + 1), oc, q, s, it)
    import os
    os.environ["LDPC_HIP_FUSED"] = "2"                # fused even where only the generic kernel exists
    try:
        dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=16, early_exit=True, want_llr=True)
    finally:
        del os.environ["LDPC_HIP_FUSED"]
    out = dec.decode(raw, s.astype(np.uint8), it, want=("bits", "iters", "synd", "status", "llr"))
    check_exact(out, ref, f"random {n}x{m} dv={dv}")
    nc = ref[3] != 0
    if nc.any():
        assert np.array_equal(out["llr"][nc].astype(np.float64), ref[4][nc])
    for inp in (raw.astype(np.int32), q.astype(np.float32), llr.astype(np.float64)):
        check_exact(dec.decode(inp, s.astype(np.uint8), it), ref, f"random {n}x{m} dv={dv} {inp.dtype}")
    # the default choice for this shape (two-pass engine where no register-cached kernel exists)
    dflt = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=16, early_exit=True)
    check_exact(dflt.decode(raw, s.astype(np.uint8), it), ref, f"random {n}x{m} dv={dv} default engine")


@pytest.mark.parametrize("shape", ["ragged_rows", "heavy_column"])
def test_q88_on_shapes_the_fused_decoder_must_turn_down(hip, oracle, tmp_path, shape):
    """Two limits of the register-cached Q8.8 kernel, both checked on the host when the decoder is made
    (the decoder then falls back to the two-pass engine and the results must still be the oracle's):
    ragged_rows  -- 992 rows of weight 2 next to 8 rows of weight 58 (one check per thread, a shape
                    the kernel is built for): every check is padded to 60 slots, ~230 pad slots share
                    each dummy LDS word, more than a dummy can absorb in one phase B and stay a pad;
    heavy_column -- one variable in 48 checks: its sum can exceed the exact range of the biased LDS
                    words (column weight <= 39)."""
    rng = np.random.default_rng(77)
    if shape == "ragged_rows":
        n, m = 1100, 1000
        H = np.zeros((m, n), np.uint8)
        for j in range(992):                                # a chain: row j checks variables j, j+1
            H[j, j] = 1
            H[j, j + 1] = 1
        for j in range(992, m):
            H[j, rng.choice(n, 58, replace=False)] = 1
        p = 0.004
    else:
        n, m = 1500, 300
        H = _random_code(rng, n, m, 3)
        H[:, 7] = 0
        H[rng.choice(m, 48, replace=False), 7] = 1
        p = 0.02
    path = str(tmp_path / "odd.alist")
    hip.alist.matrix_to_alist(H, path)
    oc = oracle.load(path)
    code = hip.Code.from_alist(path)
    x, s, llr = make_frames(oracle, oc, 24, p, 9, "bsc")
    q, raw = to_q88(llr)
    it = 12
    ref = run_oracle(lambda c, l, sy, k: oracle.dec_minsum_fixed(c, l, sy, k, None, oc.m + 1), oc, q, s, it)
    import os
    for fused in ("1", "2"):                                # default choice; generic fused kernel allowed
        os.environ["LDPC_HIP_FUSED"] = fused
        try:
            dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=64, early_exit=True)
        finally:
            del os.environ["LDPC_HIP_FUSED"]
        check_exact(dec.decode(raw, s.astype(np.uint8), it), ref, f"{shape} LDPC_HIP_FUSED={fused}")


@pytest.mark.parametrize("prec,F,it,sigma", [("q88", 1000, 15, 0.55), ("f32", 1000, 15, 0.55), ("f64", 1000, 15, 0.55),
                                             ("spa32", 1000, 15, 0.55), ("spa64", 1000, 15, 0.55), ("table", 1000, 15, 0.55),
                                             # enough frames and iterations for the two-stage early exit (>= 8 slabs, >= 16
                                             # iterations), few enough failures for its first stage to stop: the open frames
                                             # of every span are listed and decoded again as one batch
                                             ("f64", 1100, 40, 0.54), ("f32", 2200, 40, 0.54), ("spa64", 1100, 40, 0.57)])
def test_two_pass_calls_cut_into_spans(hip, oracle, prec, F, it, sigma):
    """A large call to the two-pass engine is decoded in spans of frames whose message store stays in the Infinity
    Cache (BinaryEngine::frames_per_span).  With the span squeezed to one slab (LDPC_HIP_PASS_SLABS=1), 1 000 frames
    -- a ragged last span -- with non-zero syndromes must come out as the uncut call has them (LDPC_HIP_PASS_MB=0):
    bits, iteration counts, syndromes, verdicts, posteriors; min-sum also as the oracle has them."""
    import os
    name = "1057.244.3.txt"
    oc = oracle.load(code_path(name))
    x, s, llr = make_frames(oracle, oc, F, 0.55, 31, "awgn", sigma=sigma)
    code = hip.Code.from_alist(code_path(name))
    ref = None
    if prec == "q88":
        q, inp = to_q88(llr)
        ref = run_oracle(lambda c, l, sy, k: oracle.dec_minsum_fixed(c, l, sy, k, None, oc.m + 1), oc, q, s, it)
        alg = hip.ALG.MINSUM_Q88
    elif prec == "f32":
        inp = llr.astype(np.float32)
        ref = run_oracle(oracle.dec_minsum_f32, oc, inp, s, it)
        alg = hip.ALG.MINSUM_F32
    elif prec == "f64":
        inp = llr
        ref = run_oracle(oracle.dec_minsum, oc, inp, s, it)
        alg = hip.ALG.MINSUM_F64
    elif prec == "table":
        inp = llr                                                  # (put on the Q8.14 grid by the library, as test_spa_table checks)
        alg = hip.ALG.SPA_TABLE_Q14
    else:
        inp = llr.astype(np.float32) if prec == "spa32" else llr
        alg = hip.ALG.SPA_F32 if prec == "spa32" else hip.ALG.SPA_F64
    if ref is not None:
        assert 0 < (ref[3] != 0).sum() < F                  # frames that converge and frames that do not
    os.environ["LDPC_HIP_FUSED"] = "0"                      # the two-pass engine
    try:
        decs = [hip.Decoder(code, alg, max_frames=F, early_exit=e, want_llr=True) for e in (True, False)]
    finally:
        del os.environ["LDPC_HIP_FUSED"]
    want = ("bits", "iters", "synd", "status", "llr")
    for dec in decs:
        outs = {}
        for knob, val in (("LDPC_HIP_PASS_SLABS", "1"), ("LDPC_HIP_PASS_MB", "0")):   # one slab per span; never cut
            os.environ[knob] = val
            try:
                outs[knob] = dec.decode(inp, s.astype(np.uint8), it, want=want)
            finally:
                del os.environ[knob]
            if ref is not None:
                check_exact(outs[knob], ref, f"{prec} {knob}={val}")
        nfail = int((outs["LDPC_HIP_PASS_MB"]["status"] != 0).sum())
        print(f"{prec} F={F} it={it}: {nfail} frames not converged, iterations up to {outs['LDPC_HIP_PASS_MB']['iters'].max()}")
        assert 0 < nfail < F and (it < 40 or nfail < F // 16)
        for k in want:
            assert np.array_equal(outs["LDPC_HIP_PASS_SLABS"][k], outs["LDPC_HIP_PASS_MB"][k], equal_nan=(k == "llr")), \
                f"{prec}: {k} of the cut call differs from the uncut call's"


@pytest.mark.parametrize("n,m,dv,it", [(96, 48, 3, 10), (520, 130, 3, 12), (2000, 200, 3, 8), (3000, 150, 3, 8),
                                       (1200, 1150, 3, 15), (4000, 1000, 4, 8), (777, 259, 5, 10), (6000, 300, 2, 6),
                                       (2044, 1022, 6, 10), (6000, 4000, 3, 10)])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_fused_float_minsum_on_random_shapes(hip, oracle, tmp_path, n, m, dv, it, prec):
    """The frame-resident floating-point min-sum kernel (fused_float.hip) on random codes of many
    shapes -- row weights 3 to 60 (one and two sign words), one to eight checks per thread, irregular
    LDS placement -- bit-exact against the oracle in both precisions: bits, iteration counts,
    syndromes, status, and the posterior sums of the frames that ran every iteration.  The
    profile counters prove the fused kernel (not the two-pass engine) did the work."""
    rng = np.random.default_rng(n * 11 + m)
    H = _random_code(rng, n, m, dv)
    path = str(tmp_path / "rand.alist")
    hip.alist.matrix_to_alist(H, path)
    oc = oracle.load(path)
    code = hip.Code.from_alist(path)
    F = 24
    p = {2: 0.004, 3: 0.02, 4: 0.03, 5: 0.04, 6: 0.05}[dv] * (0.5 if n / m > 5 else 1.0)
    dec = two = None
    for kind in ("awgn", "bsc"):                      # distinct magnitudes / all magnitudes tied
        x, s, llr = make_frames(oracle, oc, F, p, 6, kind, sigma=0.7 if n / m <= 5 else 0.45)
        su = s.astype(np.uint8)
        if prec == "f32":
            inp = llr.astype(np.float32)
            ref = run_oracle(oracle.dec_minsum_f32, oc, inp, s, it)
            alg = hip.ALG.MINSUM_F32
        else:
            inp = llr.astype(np.float64)
            ref = run_oracle(oracle.dec_minsum, oc, inp, s, it)
            alg = hip.ALG.MINSUM_F64
        dec = dec or hip.Decoder(code, alg, max_frames=16, want_llr=True)
        dec.profile(True)
        out = dec.decode(inp, su, it, want=("bits", "iters", "synd", "status", "llr"))
        ms, cnt = dec.profile_read()
        dec.profile(False)
        # (a frame needs q0 and sum per variable plus the compressed check states in 160 KiB of LDS;
        # where that does not fit -- 6000 x 4000 in binary64 -- q0 stays in global memory and the
        # records are packed: still the fused kernel.  Every shape here fits one way or the other.)
        size = {"f32": 4, "f64": 8}[prec]
        assert size * (2 * n + 2 * m + 300) + 16 * m < 158 * 1024 or size * (n + 2 * m + 200) + 9 * m < 158 * 1024
        assert cnt[2] > 0 and cnt[0] == 0 and cnt[1] == 0, "the fused kernel did not run"
        check_exact(out, ref, f"fused float {prec} {kind} random {n}x{m} dv={dv}")
        nc = ref[3] != 0
        if nc.any():
            assert np.array_equal(out["llr"][nc], ref[4][nc].astype(np.float32))
        # the two-pass engine (cn_variant != 0) must agree with it
        two = two or hip.Decoder(code, alg, max_frames=16, cn_variant=2)
        check_exact(two.decode(inp, su, it), ref, f"two-pass float {prec} {kind} random {n}x{m} dv={dv}")


@pytest.mark.parametrize("name,F,p,it", [("204.33.484.txt", 200, 0.06, 15), ("Frolov_1024_0.5.txt", 40, 0.07, 20),
                                         ("dvb_H_0.5.txt", 6, 0.08, 12)])
@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_spa_register_kernel_equals_two_sweep_kernels(hip, oracle, tmp_path, name, F, p, it, prec):
    """The register-resident sum-product check kernel (default for row weights <= 16, e.g. the
    DVB-S2 code) against the two-sweep kernels.  binary32: it evaluates log tanh once per row and
    keeps it where the others evaluate it twice -- same functions, same values, same order of sums:
    every output must be identical, posterior LLRs included, whichever kernel ran.  binary64 (round 3):
    the register kernel works in the tanh domain (the product of the others' tanh(a/2) in place of
    exp(sum of their log tanh), binary_kernels.hip.h spa_tanh_half64), the two-sweep kernels keep the
    sum of logarithms: the same mathematics with different roundings, so verdicts, iteration counts,
    syndromes and the words of the converged frames must be equal and the posteriors agree to 1e-9
    relative (median; NaN where the other has NaN)."""
    path = code_path(name)
    if name.startswith("dvb"):     # (the reference-style loader of the oracle needs the padded dialect)
        path = str(tmp_path / "padded.alist")
        hip.alist.normalise_alist(code_path(name), path, padded=True)
    oc = oracle.load(path)
    x, s, llr = make_frames(oracle, oc, F, p, 31, "awgn", sigma=0.9)
    inp = llr.astype(np.float32 if prec == "f32" else np.float64)
    alg = hip.ALG.SPA_F32 if prec == "f32" else hip.ALG.SPA_F64
    code = hip.Code.from_alist(code_path(name))
    want = ("bits", "iters", "synd", "status", "llr")
    outs = []
    for variant in (0, 3, 2, 1):
        dec = hip.Decoder(code, alg, max_frames=256, want_llr=True, cn_variant=variant)
        outs.append(dec.decode(inp, s.astype(np.uint8), it, want=want))
    for i, o in enumerate(outs[1:]):
        same_math = prec == "f32" or i == 0            # (variant 3 IS the default register kernel)
        for k in want:
            if k == "llr":
                if same_math:
                    assert np.array_equal(o[k], outs[0][k], equal_nan=True), k
                else:
                    a, b = o[k].astype(np.float64), outs[0][k].astype(np.float64)
                    fin = np.isfinite(a) & np.isfinite(b)
                    assert fin.mean() > 0.9
                    assert np.median(np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), 1.0)) < 1e-9
            elif k == "bits" and not same_math:
                conv = outs[0]["status"] == 0
                assert np.array_equal(o[k][conv], outs[0][k][conv]), k
            else:
                assert np.array_equal(o[k], outs[0][k]), k


@pytest.mark.parametrize("name,F,sigma,it", [("peg_16000_3_0.9.txt", 24, 0.43, 14), ("4095.737.3.txt", 60, 0.52, 14),
                                             ("32000.2240.3.105.txt", 10, 0.39, 12)])
def test_spa_f64_chunked_kernel_for_heavy_checks(hip, oracle, name, F, sigma, it):
    """Row weights 17 ... 64 (PEG-16000: 30, 4095.737: 17, 32000.2240: 43): the chunked tanh-domain check kernel
    (cn_variant 7, the default there since round 3; binary_kernels.hip.h cn_spa64_chunk_kernel) against the
    two-sweep reload kernel, which keeps the reference's sum of logarithms, and against the oracle.  Same
    mathematics, different roundings: verdicts, iteration counts and syndromes equal on every frame the oracle
    decodes without NaN, the words of its converged frames equal, posteriors to 1e-9 relative (median)."""
    oc = oracle.load(code_path(name))
    xa, sa, la = make_frames(oracle, oc, F - F // 3, 0.01, 77, "awgn", sigma=sigma)
    xb, sb, lb = make_frames(oracle, oc, F // 3, 0.01, 78, "awgn", sigma=sigma * 1.22)     # frames that do not converge
    s, llr = np.concatenate([sa, sb]), np.concatenate([la, lb])
    llr[1] *= 30.0           # saturated tanh: products of exactly 1, the reference's +inf / NaN path
    llr[2, ::9] = 0.0        # tanh(0) = 0: the reference's log(0)
    su = s.astype(np.uint8)
    ref = run_oracle(oracle.dec_spa, oc, llr, s, it)
    code = hip.Code.from_alist(code_path(name))
    want = ("bits", "iters", "synd", "status", "llr")
    outs = {}
    for variant in (0, 7, 2):
        dec = hip.Decoder(code, hip.ALG.SPA_F64, max_frames=64, want_llr=True, cn_variant=variant)
        outs[variant] = dec.decode(llr, su, it, want=want)
        dec.close()
    for k in want:                                         # 7 IS the default
        assert np.array_equal(outs[0][k], outs[7][k], equal_nan=True), k
    nan = np.isnan(ref[4]).any(axis=1)
    print(f"{name}: {nan.sum()} of {F} frames hit the reference's NaN; verdicts {np.bincount(ref[3], minlength=2)}")
    assert (~nan).sum() >= 0.5 * F and (ref[3] != 0).any() and (ref[3] == 0).any()
    for other, tag in ((outs[2], "two-sweep kernel"), ({"status": ref[3], "iters": ref[1], "synd": ref[2], "bits": ref[0],
                                                       "llr": ref[4]}, "oracle")):
        ok = ~nan
        assert np.array_equal(outs[7]["status"][ok], other["status"][ok]), tag
        assert np.array_equal(outs[7]["iters"][ok], other["iters"][ok]), tag
        assert np.array_equal(outs[7]["synd"][ok], other["synd"][ok]), tag
        conv = ok & (ref[3] == 0)
        assert np.array_equal(outs[7]["bits"][conv], other["bits"][conv]), tag
        a, b = outs[7]["llr"][ok].astype(np.float64), np.asarray(other["llr"])[ok].astype(np.float32).astype(np.float64)
        fin = np.isfinite(a) & np.isfinite(b)
        assert np.median(np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), 1.0)) < 1e-9, tag


@pytest.mark.parametrize("name,F,sigma,it", [("dvb_H_0.5.txt", 5, 0.84, 50), ("Frolov_1024_0.5.txt", 60, 0.62, 30),
                                             ("204.33.484.txt", 200, 0.55, 30)])
def test_spa_f32_follows_the_double_reference_past_binary32_saturation(hip, oracle, tmp_path, name, F, sigma, it):
    """Messages beyond |LLR| ~ 17 (where tanhf rounds to 1): the literal binary32 statement of
    logtanh2 / atanh2 turns them into inf and then NaN, the reference's double does not until
    ~37.  SPA_F32's cancellation-free evaluation must keep deciding like the double reference
    (oracle.dec_spa, test_code_B_float.c:134-199) on every frame whose double posteriors hold
    no NaN: same verdict, same word when it converged, iteration count within 2 (binary32
    rounding can move the last flip by an iteration), and no NaN of its own."""
    path = code_path(name)
    if name.startswith("dvb"):
        path = str(tmp_path / "padded.alist")
        hip.alist.normalise_alist(code_path(name), path, padded=True)
    oc = oracle.load(path)
    x, s, llr = make_frames(oracle, oc, F, 0.0, 41, "awgn", sigma=sigma)
    ref = run_oracle(oracle.dec_spa, oc, llr, s, it)
    code = hip.Code.from_alist(code_path(name))
    dec = hip.Decoder(code, hip.ALG.SPA_F32, max_frames=256, want_llr=True)
    out = dec.decode(llr.astype(np.float32), s.astype(np.uint8), it, want=("bits", "iters", "synd", "status", "llr"))
    post = ref[4]
    ok = ~np.isnan(post).any(axis=1)
    assert ok.sum() >= 0.8 * F
    assert np.abs(post[ok]).max() > 17.5, "case does not reach the binary32 saturation point"
    assert not np.isnan(out["llr"][ok]).any()
    assert np.array_equal(out["status"][ok], ref[3][ok])
    conv = ok & (ref[3] == 0)
    assert conv.sum() >= 0.5 * F
    assert np.array_equal(out["bits"][conv], ref[0][conv])
    assert (np.abs(out["iters"][conv] - ref[1][conv]) <= 2).all()
    assert (out["iters"][conv] == ref[1][conv]).mean() > 0.9


def test_high_iteration_early_exit_on_the_32000_code(hip, oracle):
    """configs[3]: 32000.2240.3.105, loop_max 200, early exit on -- frames at the edge of the
    min-sum waterfall (AWGN sigma 0.45, found by a seed search with the oracle): the Q8.8
    oracle stops two frames at 200 iterations, one after 24 and one after 133; the binary64
    one after 24 and 56.  Q8.8 and binary64 against dec() of test_code_min_sum_B_fixed.c:210-313 /
    test_code_min_sum_B.c:137-208, bit-exact, with the host's early-exit polling on and off."""
    name = "32000.2240.3.105.txt"
    oc = oracle.load(code_path(name))
    x, s, llr = make_frames(oracle, oc, 4, 0.0, 301, "awgn", sigma=0.45)
    su = s.astype(np.uint8)
    q, raw = to_q88(llr)
    code = hip.Code.from_alist(code_path(name))
    ref = run_oracle(lambda c, l, sy, k: oracle.dec_minsum_fixed(c, l, sy, k, None, oc.m + 1), oc, q, s, 200)
    assert (ref[1] > 50).sum() >= 3 and ((ref[3] == 0) & (ref[1] > 50)).any() and (ref[3] != 0).any()
    for early in (True, False):
        dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=64, early_exit=early)
        check_exact(dec.decode(raw, su, 200), ref, f"q88 32000 x 200 iterations early_exit={early}")
        dec.close()
    ref = run_oracle(oracle.dec_minsum, oc, llr, s, 200)
    assert ((ref[3] == 0) & (ref[1] > 50)).any() and (ref[3] != 0).any()
    for early in (True, False):
        dec = hip.Decoder(code, hip.ALG.MINSUM_F64, max_frames=64, early_exit=early)
        check_exact(dec.decode(llr, su, 200), ref, f"f64 32000 x 200 iterations early_exit={early}")
        dec.close()


def test_spa_f64_on_dvb_s2_across_the_waterfall(hip, oracle, tmp_path):
    """configs[2]: DVB-S2 rate 1/2 (n = 64800), SPA_F64 = the reference's own arithmetic
    (test_code_B_float.c:134-199), 50 iterations, three frames each at Eb/N0 0.6, 0.8, 1.0 and
    1.2 dB (either side of and inside the 0.8-1.0 dB waterfall, sigma^2 = 1 / (2 R 10^(dB/10))) against ora_dec_spa on the
    padded copy of the alist (the reference-style loader needs the padded dialect).
    libm differs between glibc and the device library in the last ulp, so frames are compared
    where the oracle's posteriors hold no NaN (the reference's log tanh(0) hazard): verdict, word
    and iteration count equal on every such frame that converges; frames that do not converge
    (below the waterfall) must agree on the verdict and on >= 99.9 % of the bits."""
    path = str(tmp_path / "padded.alist")
    hip.alist.normalise_alist(code_path("dvb_H_0.5.txt"), path, padded=True)
    oc = oracle.load(path)
    code = hip.Code.from_alist(code_path("dvb_H_0.5.txt"))
    dec = hip.Decoder(code, hip.ALG.SPA_F64, max_frames=64, want_llr=True)
    seen_conv = seen_fail = 0
    for db, seed in ((0.6, 51), (0.8, 53), (1.0, 54), (1.2, 52)):
        sigma = (1.0 / (2 * 0.5 * 10 ** (db / 10))) ** 0.5
        x, s, llr = make_frames(oracle, oc, 3, 0.0, seed, "awgn", sigma=sigma)
        ref = run_oracle(oracle.dec_spa, oc, llr, s, 50)
        out = dec.decode(llr, s.astype(np.uint8), 50, want=("bits", "iters", "synd", "status", "llr"))
        ok = ~np.isnan(ref[4]).any(axis=1)
        print(f"dvb spa_f64 {db} dB: oracle (rc, iters) {list(zip(ref[3], ref[1]))} gpu {list(zip(out['status'], out['iters']))} nan-free {ok}")
        for f in np.where(ok)[0]:
            assert out["status"][f] == ref[3][f], (db, f)
            if ref[3][f] == 0:
                seen_conv += 1
                assert out["iters"][f] == ref[1][f] and np.array_equal(out["bits"][f], ref[0][f]), (db, f)
                assert np.array_equal(out["synd"][f], ref[2][f])
            else:
                seen_fail += 1
                assert (out["bits"][f] != ref[0][f]).mean() <= 1e-3, (db, f)
    assert seen_conv >= 2 and seen_fail >= 2, (seen_conv, seen_fail)


@pytest.mark.parametrize("alg,F,db,it", [("MINSUM_F64", 1200, 2.0, 30), ("SPA_F32", 2100, 1.05, 40), ("MINSUM_Q88", 2600, 2.0, 12)])
def test_default_spans_on_dvb_s2_equal_the_uncut_call(hip, alg, F, db, it):
    """The span sizes the engine picks by itself, at sizes where they matter: DVB-S2 rate 1/2 on the two-pass engine
    (LDPC_HIP_FUSED=0, LDPC_HIP_CHAIN=0), a slab of 232 MB, spans of two slabs with a ragged tail, frames on either side
    of the waterfall so that the two-stage early exit (>= 8 slabs, >= 16 iterations) stops spans early and decodes their
    open frames as one batch.  The uncut call (LDPC_HIP_PASS_MB=0) is the reference point; it is checked against the
    oracle elsewhere (test_minsum_*, test_spa_*): every output of the cut call must equal it, bit for bit."""
    import os
    code = hip.Code.from_alist(code_path("dvb_H_0.5.txt"))
    n, m = code.n, code.m
    rng = np.random.default_rng(101)
    sigma = (1.0 / (2 * 0.5 * 10 ** (db / 10))) ** 0.5
    # all-zero codeword; one frame in 25 at twice the noise power (they fail: fewer than the 1/16 at which a span's first stage stops), zero syndromes
    sig = np.where(np.arange(F) % 25 == 3, sigma * 1.41, sigma)[:, None]
    y = 1.0 + rng.normal(0.0, 1.0, size=(F, n)) * sig
    llr = 2.0 * y / sig ** 2
    if alg == "MINSUM_Q88":
        _, inp = to_q88(llr)
    else:
        inp = llr.astype(np.float32) if alg.endswith("F32") else llr
    os.environ["LDPC_HIP_FUSED"] = "0"
    os.environ["LDPC_HIP_CHAIN"] = "0"
    try:
        dec = hip.Decoder(code, getattr(hip.ALG, alg), max_frames=F, early_exit=True, want_llr=True)
    finally:
        del os.environ["LDPC_HIP_FUSED"], os.environ["LDPC_HIP_CHAIN"]
    want = ("bits", "iters", "synd", "status", "llr")
    cut = dec.decode(inp, None, it, want=want)
    os.environ["LDPC_HIP_PASS_MB"] = "0"
    try:
        whole = dec.decode(inp, None, it, want=want)
    finally:
        del os.environ["LDPC_HIP_PASS_MB"]
    nfail = int((whole["status"] != 0).sum())
    print(f"dvb {alg}: {nfail} of {F} frames not converged, iterations {whole['iters'].min()} .. {whole['iters'].max()}")
    assert 0 < nfail < F
    for k in want:
        assert np.array_equal(cut[k], whole[k], equal_nan=(k == "llr")), f"{alg}: {k} of the cut call differs from the uncut call's"
