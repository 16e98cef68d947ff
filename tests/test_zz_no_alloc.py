"""ldpc_decode_batch_device_noalloc (the allocation-free entry point, include/ldpc_hip.h) against
ldpc_decode_batch_device on the same device buffers, through the C ABI: two calls in a row, the
second with fewer frames (the static scratch keeps the first call's contents), for a
frame-resident decoder (scratch = frame queue head) and for two-pass decoders (scratch = message
store and per-frame state); sizes beyond the 32 MiB are refused by code.  Bit-exact: both entry
points run the same kernels."""
import numpy as np
import pytest

from conftest import code_path, make_frames, to_q88

pytestmark = pytest.mark.gpu


class _Hip:
    """Device buffers through the HIP runtime the library itself is linked to (no torch in the
    process: one HIP runtime per process)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.rt.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.rt.hipStreamDestroy.argtypes = [C.c_void_p]
        self.held = []

    def free_bytes(self):
        """Free device memory as the runtime sees it (after everything queued has finished)."""
        fr, tot = self.C.c_size_t(), self.C.c_size_t()
        assert self.rt.hipDeviceSynchronize() == 0
        assert self.rt.hipMemGetInfo(self.C.byref(fr), self.C.byref(tot)) == 0
        return fr.value

    def stream(self):
        st = self.C.c_void_p()
        assert self.rt.hipStreamCreate(self.C.byref(st)) == 0
        return st

    def up(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), max(1, arr.nbytes)) == 0
        assert self.rt.hipMemcpy(p, arr.ctypes.data_as(self.C.c_void_p), arr.nbytes, 1) == 0
        self.held.append(p)
        return p.value

    def filled(self, shape, dtype, byte):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), max(1, n)) == 0
        assert self.rt.hipMemset(p, byte, n) == 0
        self.held.append(p)
        return p.value

    def down(self, ptr, shape, dtype):
        out = np.empty(shape, dtype)
        assert self.rt.hipDeviceSynchronize() == 0
        assert self.rt.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), self.C.c_void_p(ptr), out.nbytes, 2) == 0
        return out

    def free(self):
        for p in self.held:
            self.rt.hipFree(p)
        self.held = []


def _run(hip, dec, code, inp, synd, it, noalloc, want_llr=False):
    g = _Hip()
    F = inp.shape[0]
    d_in, d_sy = g.up(inp), g.up(synd)
    bits = g.filled((F, code.n), np.uint8, 7)
    sout = g.filled((F, code.m), np.uint8, 7)
    iters = g.filled((F,), np.int32, 0xF7)
    status = g.filled((F,), np.int32, 0xF7)
    post = g.filled((F, code.n), np.float32, 0) if want_llr else None
    dt = {np.dtype(np.int16): hip.DT.I16, np.dtype(np.float32): hip.DT.F32, np.dtype(np.float64): hip.DT.F64,
          np.dtype(np.int32): hip.DT.I32}[inp.dtype]
    try:
        before = g.free_bytes()
        dec.decode_device(F, d_in, dt, it, d_sy, bits, iters, sout, status, post, None, noalloc=noalloc)
        after = g.free_bytes()
        out = dict(bits=g.down(bits, (F, code.n), np.uint8), synd=g.down(sout, (F, code.m), np.uint8),
                   iters=g.down(iters, (F,), np.int32), status=g.down(status, (F,), np.int32))
        if want_llr:
            out["llr"] = g.down(post, (F, code.n), np.float32)
        out["_allocated"] = before - after      # device bytes the call itself took from the runtime
    finally:
        g.free()
    return out


CASES = [
    # code, algorithm, decoder options, frames of the first / second call, channel
    ("1057.244.3.txt", "MINSUM_Q88", {}, 300, 41, 0.012),                       # frame-resident, int16 LLRs
    ("204.33.484.txt", "MINSUM_F32", {}, 200, 37, 0.05),                       # frame-resident binary32
    ("204.33.484.txt", "MINSUM_Q88", {"cn_variant": 5}, 700, 129, 0.05),       # two-pass, int16 messages, 2 slabs then 1
    ("204.33.484.txt", "SPA_F32", {"want_llr": True}, 300, 50, 0.05),          # two-pass binary32 + posterior LLRs
    ("204.33.484.txt", "SPA_TABLE_Q14", {"early_exit": True}, 260, 64, 0.04),  # two-pass int32, host early-exit polls
    # staircase frame-resident kernel: its per-workgroup rows (R * block * 10 B = 320 KiB for this code) come out of
    # the scratch, so the grid is capped at the 102 workgroups whose rows fit 32 MiB
    ("dvb_H_0.5.txt", "MINSUM_Q88", {}, 300, 9, 0.05),
    # wide-row binary32 kernel (q0 in registers: the queue head is all it takes from the scratch)
    ("32000.2240.3.105.txt", "MINSUM_F32", {}, 150, 9, 0.0005),
    # staircase binary32 decoder: 28 bytes per check and workgroup (896 KiB for this code): 36 workgroups fit
    ("dvb_H_0.5.txt", "MINSUM_F32", {}, 100, 9, 0.04),
]


@pytest.mark.parametrize("name,alg,opts,f1,f2,p", CASES)
def test_noalloc_equals_the_allocating_entry_point_twice_in_a_row(hip, oracle, tmp_path, name, alg, opts, f1, f2, p):
    path = code_path(name)
    if name.startswith("dvb"):     # (the reference-style loader of the oracle needs the padded dialect)
        path = str(tmp_path / "padded.alist")
        hip.alist.normalise_alist(code_path(name), path, padded=True)
    oc = oracle.load(path)
    code = hip.Code.from_alist(code_path(name))
    o = dict(early_exit=False)
    o.update(opts)
    dec = hip.Decoder(code, getattr(hip.ALG, alg), max_frames=1024, **o)
    for F, seed in ((f1, 11), (f2, 12)):
        x, s, llr = make_frames(oracle, oc, F, p, seed, "bsc")
        if alg == "MINSUM_Q88":
            inp = to_q88(llr)[1]
        elif alg == "SPA_TABLE_Q14":
            inp = oracle.float2fix14(llr).astype(np.int32)
        else:
            inp = llr.astype(np.float32)
        sy = s.astype(np.uint8)
        want_llr = bool(o.get("want_llr"))
        ref = _run(hip, dec, code, inp, sy, 20, False, want_llr)
        got = _run(hip, dec, code, inp, sy, 20, True, want_llr)
        assert (ref["iters"] >= 0).all() and set(np.unique(ref["status"])) <= {0, 1}
        if F == f1:
            assert (ref["status"] == 0).any(), "the first call should hold frames that converge"
        for k in ref:       # (the float sum-product keeps the reference's NaN hazards: NaN == NaN here)
            if k != "_allocated":
                assert np.array_equal(ref[k], got[k], equal_nan=(k == "llr")), (name, alg, F, k)
        # the point of the entry: the call takes no device memory from the runtime (the allocating entry may,
        # on its first call; by the second call of either kind everything lazily allocated exists)
        if F == f2:
            assert got["_allocated"] == 0, f"the allocation-free call allocated {got['_allocated']} bytes"
    dec.close()


def test_noalloc_refuses_what_does_not_fit_and_stays_usable(hip, oracle):
    name = "Frolov_1024_0.5.txt"
    oc = oracle.load(code_path(name))
    code = hip.Code.from_alist(code_path(name))
    # two-pass int16 engine: one slab of 512 frames of this code needs ~10 MB, four or more do not fit 32 MiB
    dec = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=4096, early_exit=False, cn_variant=5)
    F = 3585
    x, s, llr = make_frames(oracle, oc, F, 0.03, 5, "bsc")
    raw = to_q88(llr)[1]
    sy = s.astype(np.uint8)
    with pytest.raises(hip.LdpcError) as e:
        _run(hip, dec, code, raw, sy, 10, True)
    assert "static scratch" in str(e.value) and e.value.code == -4
    with pytest.raises(hip.LdpcError) as e:                                   # poll slots were sized at create
        _run(hip, dec, code, raw[:8], sy[:8], 1000, True)
    assert "iterations" in str(e.value)
    ref = _run(hip, dec, code, raw[:1000], sy[:1000], 10, False)
    got = _run(hip, dec, code, raw[:1000], sy[:1000], 10, True)
    for k in ref:
        if k != "_allocated":
            assert np.array_equal(ref[k], got[k]), k
    # non-binary decoders are not served
    g = hip.Code.from_alist(code_path("q4.sp.9000.6000.4500.1"))
    gd = hip.Decoder(g, hip.ALG.GF2P_WHT, max_frames=2)
    mem = _Hip()
    z = mem.filled((1, g.n, 4), np.int32, 0)
    with pytest.raises(hip.LdpcError) as e:
        gd.decode_device(1, z, hip.DT.I32, 1, noalloc=True)
    assert e.value.code == -6
    mem.free()
    dec.close()
    gd.close()


def test_noalloc_refuses_overlapping_calls(hip, oracle):
    """Every call on a GPU works in that GPU's one static array, so two calls must not overlap: a call that
    arrives while another is on the host (two host threads), or while the previous call's kernels are still
    running on ANOTHER stream, is refused with LDPC_ERR_ARG instead of racing; the same stream, or any
    stream once that work is done, is served."""
    import threading
    import time
    name = "1057.244.3.txt"
    oc = oracle.load(code_path(name))
    code = hip.Code.from_alist(code_path(name))
    F = 2048
    x, s, llr = make_frames(oracle, oc, F, 0.09, 3, "bsc")            # far above threshold: nothing converges
    raw = to_q88(llr)[1]
    g = _Hip()
    d_in = g.up(raw)
    bitsA, statA = g.filled((F, code.n), np.uint8, 0), g.filled((F,), np.int32, 0)
    bitsB, statB = g.filled((F, code.n), np.uint8, 0), g.filled((F,), np.int32, 0)
    # (a) two host threads: A's call polls the host every iteration (early exit on, two-pass engine) and
    # stays in the library for 254 iterations; B knocks meanwhile
    decA = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=F, early_exit=True, cn_variant=5)
    decB = hip.Decoder(code, hip.ALG.MINSUM_Q88, max_frames=64, early_exit=False)
    started, finished = threading.Event(), threading.Event()
    errs = []

    def a_side():
        try:
            started.set()
            decA.decode_device(F, d_in, hip.DT.I16, 254, None, bitsA, None, None, statA, None, None, noalloc=True)
        except Exception as e:          # noqa: BLE001 -- reported below
            errs.append(e)
        finally:
            finished.set()
    th = threading.Thread(target=a_side)
    th.start()
    started.wait()
    time.sleep(0.002)
    refused = served = 0
    while not finished.is_set():
        try:
            decB.decode_device(8, d_in, hip.DT.I16, 2, None, bitsB, None, None, statB, None, None, noalloc=True)
            served += 1
        except hip.LdpcError as e:
            assert e.code == -4 and "another call" in str(e)
            refused += 1
    th.join()
    assert not errs, errs
    assert refused > 0, f"no call of thread B met thread A's ({served} served)"
    assert g.rt.hipDeviceSynchronize() == 0
    refA = g.down(statA, (F,), np.int32)
    assert (refA == 1).all()
    # (b) two streams: an asynchronous call (early exit off: the call returns while its kernel runs) on one
    # stream, then a call on another stream
    peg = hip.Code.from_alist(code_path("peg_16000_3_0.9.txt"))
    Fp = 4096
    rng = np.random.default_rng(1)
    big = np.where(rng.random((Fp, peg.n)) < 0.02, -996, 996).astype(np.int16)
    d_big = g.up(big)
    bitsP, statP = g.filled((Fp, peg.n), np.uint8, 0), g.filled((Fp,), np.int32, 0)
    decP = hip.Decoder(peg, hip.ALG.MINSUM_Q88, max_frames=Fp, early_exit=False)
    s1, s2 = g.stream(), g.stream()
    decP.decode_device(Fp, d_big, hip.DT.I16, 200, None, bitsP, None, None, statP, None, s1.value, noalloc=True)
    decP.decode_device(64, d_big, hip.DT.I16, 200, None, bitsP, None, None, statP, None, s1.value, noalloc=True)   # same stream: ordered
    with pytest.raises(hip.LdpcError) as e:
        decP.decode_device(64, d_big, hip.DT.I16, 10, None, bitsP, None, None, statP, None, s2.value, noalloc=True)
    assert e.value.code == -4 and "another stream" in str(e.value)
    assert g.rt.hipDeviceSynchronize() == 0
    decP.decode_device(64, d_big, hip.DT.I16, 10, None, bitsP, None, None, statP, None, s2.value, noalloc=True)  # idle now: served
    assert g.rt.hipDeviceSynchronize() == 0
    assert (g.down(statP, (64,), np.int32) == 1).all()
    g.rt.hipStreamDestroy(s1)
    g.rt.hipStreamDestroy(s2)
    g.free()
    for d in (decA, decB, decP):
        d.close()
