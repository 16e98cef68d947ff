"""oracle/mc_frontend.py -- CPU restatement (numpy) of the device Monte-Carlo front-end.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): imported by tests/ alone, never by the
product.  It restates, frame by frame, what ldpc_amd/csrc/mc_frontend.hip generates:

  source word   x[j] = rand() & 1                       test_code_min_sum_B.c:308-311
  syndrome      s = enc(x), s[j] = XOR of x over row j  test_code_min_sum_B.c:69-79
  bsc()         exactly round(n p) distinct flips found by rejection, p' = flips / n,
                q0[i] = (1 - 2 y[i]) log((1 - p') / p') test_code_min_sum_B.c:41-67
  float_to_fix  round half away, clip to [-128, 127]    test_code_min_sum_B_fixed.c:43-53
  float2fix     Q8.14, round half away                  test_code_B.c:36-43

The reference draws from libc rand() seeded with time(NULL) (:289), so there is no
reference stream to match: PARITY UNPINNED for the random numbers themselves.  What is
pinned is the channel's law (number of flips, LLR values, quantisation, syndrome), which
the tests check against this restatement bit for bit, using the library's own
counter-based generator: draw i of stream (seed, frame, s) = mix64(key + (i+1) GOLDEN),
key = mix64(seed ^ mix64(4 frame + s + GOLDEN)), mix64 = the splitmix64 finaliser.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
BSC_EXACT, BSC_IID, AWGN = range(3)
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


def generate(n, row_ptr, col_idx, nframes, channel, param, dtype, seed, first_frame=0, source=SRC_RANDOM):
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
