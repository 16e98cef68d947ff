"""oracle/mc_frontend.py -- CPU restatement (numpy) of the device Monte-Carlo front-end.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): imported by tests/ alone, never by the
product.  It restates, frame by frame, what ldpc_amd/csrc/mc_frontend.hip generates:

  source word   x[j] = rand() & 1                       test_code_min_sum_B.c:308-311
  syndrome      s = enc(x), s[j] = XOR of x over row j  test_code_min_sum_B.c:69-79
  bsc()         exactly round(n p) distinct flips found by rejection, p' = flips / n,
                q0[i] = (1 - 2 y[i]) log((1 - p') / p') test_code_min_sum_B.c:41-67
  float_to_fix  round half away, clip to [-128, 127]    test_code_min_sum_B_fixed.c:43-53
  float2fix     Q8.14, round half away                  test_code_B.c:36-43
  QLC channel   Gray map, percent-granular inverse CDF  test_code_B_MSDP.c:411-440
                make_p_sent_given_rec_T, bit LLRs       test_code_B_MSDP.c:297-395

The reference draws from libc rand() seeded with time(NULL) (:289), so there is no
reference stream to match: PARITY UNPINNED for the random numbers themselves.  What is
pinned is the channel's law (number of flips, LLR values, quantisation, syndrome), which
the tests check against this restatement bit for bit, using the library's own
counter-based generator: draw i of stream (seed, frame, s) = mix64(key + (i+1) GOLDEN),
key = mix64(seed ^ mix64(4 frame + s + GOLDEN)), mix64 = the splitmix64 finaliser.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
BSC_EXACT, BSC_IID, AWGN, QLC = range(4)
SRC_RANDOM, SRC_FIXED, SRC_ZERO = range(3)
F64, F32, I16, I32 = range(4)


def mix64(z):
    z = np.asarray(z, np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def stream_key(seed, frame, stream):
    with np.errstate(over="ignore"):
        return mix64(np.uint64(seed) ^ mix64(np.uint64(frame) * np.uint64(4) + np.uint64(stream) + GOLDEN))


def draw(key, i):
    with np.errstate(over="ignore"):
        return mix64(key + (np.asarray(i, np.uint64) + np.uint64(1)) * GOLDEN)


def float_to_fix_raw(v):
    r = np.sign(v) * np.floor(np.abs(v) * 256.0 + 0.5)         # C round(): half away from zero
    return np.clip(r, -128.0 * 256.0, 127.0 * 256.0).astype(np.int16)


def float2fix14(v):
    return (np.sign(v) * np.floor(np.abs(v) * 16384.0 + 0.5)).astype(np.int32)


def source_word(n, seed, frame, source):
    if source == SRC_ZERO:
        return np.zeros(n, np.uint8)
    nw = (n + 31) // 32
    key = stream_key(seed, np.uint64(0xFFFFFFFFFFFFFFFF) if source == SRC_FIXED else frame, 0)
    words = (draw(key, np.arange(nw)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    bits = ((words[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.uint8).reshape(-1)
    return bits[:n]


def bsc_exact_pattern(n, p, seed, frame):
    flips = min(int(n * p + 0.5), n)
    key = stream_key(seed, frame, 1)
    err = np.zeros(n, np.uint8)
    have, i0 = 0, 0
    while have < flips:
        cnt = max(64, 2 * (flips - have))
        u = draw(key, np.arange(i0, i0 + cnt))
        pos = (((u >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)
        for q in pos:                      # sequential rejection, like the reference's while loop
            if have == flips:
                break
            if not err[q]:
                err[q] = 1
                have += 1
        i0 += cnt
    pp = flips / n
    with np.errstate(divide="ignore"):
        mag = np.log((1.0 - pp) / pp) if flips else np.inf
    return err, mag


GREY_CODE_INV = [6, 5, 7, 14, 9, 12, 8, 13, 3, 4, 2, 15, 10, 11, 1, 0]            # test_code_B_MSDP.c:23
SYM_SETS = [[0, 4, 5, 11, 12, 13, 14, 15], [0, 1, 2, 7, 8, 13, 14, 15],         # low, mid (:27, :26)
            [0, 1, 8, 9, 10, 11, 12, 13], [0, 1, 2, 3, 4, 10, 11, 15]]           # up, top (:25, :24)


def qlc_llr_table(conf):
    """make_p_sent_given_rec_T() (:297-323) then assign_llr_one_sym() (:339-395): [rows][4],
    sums in the reference's order."""
    rows, Q = conf.shape
    P_x = 1 / (1.0 * Q)
    tab = np.zeros((rows, 4))
    for i in range(rows):
        P_y = 0.0
        for j in range(Q):
            P_y += conf[i, j] * P_x
        T = [P_x * conf[i, j] / P_y if P_y else 0.0 for j in range(Q)]
        for b in range(4):
            pr1 = 0.0
            for j in SYM_SETS[b]:
                pr1 += T[j]
            with np.errstate(invalid="ignore"):       # a sum one ulp above 1 gives log(<0) = NaN in the reference too
                tab[i, b] = -100.0 if pr1 == 1 else 100.0 if pr1 == 0 else np.log((1.0 - pr1) / pr1)
    return tab


def qlc_channel(x, conf, tab, key):
    """channel(), :411-440, with rand() % 101 replaced by the frame's stream."""
    rows = conf.shape[0]
    cells = len(x) // 4
    d = np.zeros(len(x))
    sel = ((draw(key, np.arange(cells)) >> np.uint64(32)) * np.uint64(101)) >> np.uint64(32)
    for c in range(cells):
        symbol = GREY_CODE_INV[(int(x[4 * c + 3]) << 3) + (int(x[4 * c + 2]) << 2) + (int(x[4 * c + 1]) << 1) + int(x[4 * c])]
        temp, rec, rs = 0.0, 0, int(sel[c])
        if not rs:
            while rec < rows:
                rec += 1
                if conf[rec - 1, symbol]:
                    break
        while temp < rs and rec != rows:
            temp += 100 * conf[rec, symbol]
            rec += 1
        rec = rec - 1 if rec > 0 else 0
        d[4 * c:4 * c + 4] = tab[rec]
    return d


def generate(n, row_ptr, col_idx, nframes, channel, param, dtype, seed, first_frame=0, source=SRC_RANDOM, conf=None):
    """-> x [F][n] u8, llr [F][n] of dtype, synd [F][m] u8, llr as float64 before quantisation."""
    m = len(row_ptr) - 1
    x = np.zeros((nframes, n), np.uint8)
    d = np.zeros((nframes, n), np.float64)
    s = np.zeros((nframes, m), np.uint8)
    for f in range(nframes):
        frame = first_frame + f
        x[f] = source_word(n, seed, frame, source)
        key = stream_key(seed, frame, 1)
        if channel == BSC_EXACT:
            err, mag = bsc_exact_pattern(n, param, seed, frame)
            d[f] = (1 - 2 * (x[f] ^ err).astype(np.float64)) * mag
        elif channel == QLC:
            d[f] = qlc_channel(x[f], conf, qlc_llr_table(conf), key)
        elif channel == BSC_IID:
            thr = np.uint64(0xFFFFFFFFFFFFFFFF) if param >= 1 else np.uint64(int(param * 18446744073709551616.0))
            err = (draw(key, np.arange(n)) < thr).astype(np.uint8)
            d[f] = (1 - 2 * (x[f] ^ err).astype(np.float64)) * np.log((1.0 - param) / param)
        else:
            i = np.arange(n, dtype=np.uint64)
            u1 = ((draw(key, 2 * i) >> np.uint64(11)).astype(np.float64) + 1.0) / 9007199254740992.0
            u2 = (draw(key, 2 * i + 1) >> np.uint64(11)).astype(np.float64) / 9007199254740992.0
            g = np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586476925 * u2)
            y = (1 - 2 * x[f].astype(np.float64)) + param * g
            d[f] = 2.0 * y / (param * param)
        # enc(): rows are never empty in an alist, so reduceat over the row starts is the row sum
        s[f] = np.add.reduceat(x[f, col_idx].astype(np.int64), row_ptr[:-1]) & 1
    if dtype == F64:
        llr = d.copy()
    elif dtype == F32:
        llr = d.astype(np.float32)
    elif dtype == I16:
        llr = float_to_fix_raw(np.where(np.isinf(d), np.sign(d) * 1e9, d))
    else:
        llr = float2fix14(d)
    return x, llr, s, d
