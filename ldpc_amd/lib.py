"""ctypes binding of libldpc_hip.so (C ABI: include/ldpc_hip.h).

Mirrors the header one to one; numpy arrays in, numpy arrays out.  The library
is loaded lazily and a missing library is an error (no fallback path).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class LdpcError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libldpc_hip error {code}: {text}")
        self.code = code


class ALG:
    MINSUM_F32, MINSUM_F64, MINSUM_Q88, SPA_F32, SPA_F64, SPA_TABLE_Q14, GF2P_WHT, GFQ_DIRECT = range(8)


class DT:
    F64, F32, I16, I32 = range(4)


_NP2DT = {np.dtype(np.float64): DT.F64, np.dtype(np.float32): DT.F32,
          np.dtype(np.int16): DT.I16, np.dtype(np.int32): DT.I32}


class _Opts(C.Structure):
    _fields_ = [("alg", C.c_int32), ("device", C.c_int32), ("max_frames", C.c_int32),
                ("early_exit", C.c_int32), ("want_llr", C.c_int32), ("cn_variant", C.c_int32),
                ("layer_checks", C.c_int32), ("reserved", C.c_int32 * 9)]


class _McOpts(C.Structure):
    _fields_ = [("channel", C.c_int32), ("source", C.c_int32), ("llr_dtype", C.c_int32), ("max_iter", C.c_int32),
                ("param", C.c_double), ("seed", C.c_uint64), ("first_frame", C.c_uint64),
                ("reserved", C.c_int32 * 8)]


class CH:
    BSC_EXACT, BSC_IID, AWGN, QLC = range(4)


class SRC:
    RANDOM, FIXED, ZERO = range(3)


def lib_path():
    return os.environ.get("LDPC_HIP_LIB", os.path.join(_HERE, "libldpc_hip.so"))


_lib = None


def load_library():
    """Loads libldpc_hip.so; raises OSError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise OSError(f"{path} not found: build it with `make -C ldpc_amd/csrc` "
                      "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(path)
    vp, ip, cp = C.c_void_p, C.c_int, C.c_char_p
    L.ldpc_strerror.restype = cp
    L.ldpc_strerror.argtypes = [ip]
    L.ldpc_last_error.restype = cp
    L.ldpc_version.restype = cp
    L.ldpc_code_load_alist.argtypes = [cp, C.POINTER(vp)]
    L.ldpc_code_from_csr.argtypes = [ip, ip, ip, vp, vp, vp, C.POINTER(vp)]
    L.ldpc_code_free.argtypes = [vp]
    L.ldpc_code_free.restype = None
    for f in ("n", "m", "q", "edges", "cmax", "rmax"):
        getattr(L, "ldpc_code_" + f).argtypes = [vp]
    L.ldpc_code_get_csr.argtypes = [vp, vp, vp, vp]
    L.ldpc_code_get_csc.argtypes = [vp, vp, vp]
    L.ldpc_code_syndrome.argtypes = [vp, vp, vp]
    L.ldpc_decoder_create.argtypes = [vp, C.POINTER(_Opts), C.POINTER(vp)]
    L.ldpc_decoder_free.argtypes = [vp]
    L.ldpc_decoder_free.restype = None
    L.ldpc_decoder_max_frames.argtypes = [vp]
    L.ldpc_decoder_last_iterations.argtypes = [vp]
    L.ldpc_decode_batch.argtypes = [vp, ip, vp, ip, vp, ip, vp, vp, vp, vp, vp]
    L.ldpc_decode_batch_device.argtypes = [vp, ip, vp, ip, vp, ip, vp, vp, vp, vp, vp, vp]
    L.ldpc_decode_batch_device_noalloc.argtypes = [vp, ip, vp, ip, vp, ip, vp, vp, vp, vp, vp, vp]
    L.ldpc_decode_trace_q88.argtypes = [vp, vp, ip, vp, ip, vp, vp, vp, vp, cp, ip]
    L.ldpc_code_lds_schedule_stats.argtypes = [vp, ip, ip, vp]
    L.ldpc_code_vpass_place_stats.argtypes = [vp, ip, vp]
    L.ldpc_decoder_profile.argtypes = [vp, ip]
    L.ldpc_decoder_profile_read.argtypes = [vp, vp, vp]
    L.ldpc_decoder_profile_frames.argtypes = [vp, vp]
    L.ldpc_mc_create.argtypes = [vp, vp, ip, C.POINTER(vp)]
    L.ldpc_mc_free.argtypes = [vp]
    L.ldpc_mc_free.restype = None
    L.ldpc_mc_run.argtypes = [vp, C.POINTER(_McOpts), C.c_int64, vp]
    L.ldpc_mc_generate.argtypes = [vp, C.POINTER(_McOpts), ip, vp, vp, vp]
    L.ldpc_mc_last_pass.argtypes = [vp, ip, vp, vp, vp, vp]
    L.ldpc_mc_set_qlc_channel.argtypes = [vp, vp, ip]
    L.ldpc_gfmc_create.argtypes = [vp, vp, ip, C.POINTER(vp)]
    L.ldpc_gfmc_free.argtypes = [vp]
    L.ldpc_gfmc_free.restype = None
    L.ldpc_gfmc_set_channel.argtypes = [vp, vp, ip]
    L.ldpc_gfmc_run.argtypes = [vp, ip, C.c_int64, C.c_uint64, C.c_uint64, vp]
    L.ldpc_gfmc_generate.argtypes = [vp, C.c_uint64, C.c_uint64, ip, vp, vp, vp, vp, vp]
    L.ldpc_gfmc_last_pass.argtypes = [vp, ip, vp, vp, vp, vp]
    L.ldpc_wht_tables_selfcheck.argtypes = [vp]
    L.ldpc_wht_tables_selfcheck.restype = C.c_int64
    L.ldpc_spa_tables_selfcheck.argtypes = [vp]
    L.ldpc_spa_tables_selfcheck.restype = C.c_int64
    L.ldpc_set_seed.argtypes = [C.c_uint64]
    L.ldpc_set_seed.restype = None
    dp, intp = C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.test_code_min_sum_B.argtypes = [cp, ip, intp, intp, intp, dp, dp]
    L.test_code_min_sum_B_cw_noise_gen.argtypes = [cp, ip, ip, C.c_double, dp]
    L.test_code_min_sum_B_fixed.argtypes = [cp, ip, intp, intp, intp, dp, dp, cp]
    L.test_code_min_sum_B_fixed_cw_noise_gen.argtypes = [cp, ip, ip, C.c_double, dp]
    L.test_code_B_float.argtypes = [cp, ip, intp, intp, intp, dp, dp]
    L.test_code_B_float_cw_noise_gen.argtypes = [cp, ip, ip, C.c_double, dp]
    L.test_code_B.argtypes = [ip, ip, C.c_double, cp, dp]
    L.GFq_LDPC.argtypes = [cp, ip, ip, ip, ip, ip, dp, dp]
    L.GFq_LDPC_experiment.argtypes = [cp, ip, dp, C.c_uint64, ip, vp, vp, vp, vp, vp]
    L.GFq_LDPC_graph.argtypes = [ip, ip, ip, ip, ip, ip, dp, ip, ip, ip, ip, intp, intp, intp, intp, intp, intp, dp]
    L.test_code_B_MSDP.argtypes = [cp, ip, ip, ip, ip, dp, dp]
    L.GFq_LDPC_NTT.argtypes = [ip, cp, C.c_double, cp, vp, vp]
    L.GFq_LDPC_NTT_experiment.argtypes = [cp, ip, cp, C.c_double, vp, vp, vp, vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        L = load_library()
        text = L.ldpc_last_error().decode() or L.ldpc_strerror(rc).decode()
        raise LdpcError(rc, text)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Code:
    """A parsed H matrix (replaces initdec() and its globals)."""

    def __init__(self, handle):
        self._h = handle
        L = load_library()
        self.n, self.m, self.q = L.ldpc_code_n(handle), L.ldpc_code_m(handle), L.ldpc_code_q(handle)
        self.E = L.ldpc_code_edges(handle)
        self.cmax, self.rmax = L.ldpc_code_cmax(handle), L.ldpc_code_rmax(handle)

    @classmethod
    def from_alist(cls, path):
        L = load_library()
        h = C.c_void_p()
        check(L.ldpc_code_load_alist(os.fsencode(path), C.byref(h)))
        return cls(h)

    @classmethod
    def from_csr(cls, n, m, row_ptr, col_idx, hval=None, q=2):
        L = load_library()
        rp = np.ascontiguousarray(row_ptr, np.int32)
        ci = np.ascontiguousarray(col_idx, np.int32)
        hv = None if hval is None else np.ascontiguousarray(hval, np.int32)
        h = C.c_void_p()
        check(L.ldpc_code_from_csr(n, m, q, _ptr(rp), _ptr(ci), _ptr(hv), C.byref(h)))
        return cls(h)

    @classmethod
    def from_dense(cls, H, q=2):
        """Dense H[m][n] (the reference's Python prototype passes H like this)."""
        H = np.asarray(H)
        m, n = H.shape
        rows, cols = np.nonzero(H)          # row-major order = ascending column in each row
        row_ptr = np.zeros(m + 1, np.int32)
        np.add.at(row_ptr, rows + 1, 1)
        row_ptr = np.cumsum(row_ptr).astype(np.int32)
        hval = H[rows, cols].astype(np.int32) if q > 2 else None
        return cls.from_csr(n, m, row_ptr, cols.astype(np.int32), hval, q)

    def csr(self):
        rp = np.zeros(self.m + 1, np.int32)
        ci = np.zeros(self.E, np.int32)
        hv = np.zeros(self.E, np.int32)
        check(load_library().ldpc_code_get_csr(self._h, _ptr(rp), _ptr(ci), _ptr(hv)))
        return rp, ci, hv

    def csc(self):
        cp = np.zeros(self.n + 1, np.int32)
        eo = np.zeros(self.E, np.int32)
        check(load_library().ldpc_code_get_csc(self._h, _ptr(cp), _ptr(eo)))
        return cp, eo

    def syndrome(self, x):
        """enc(): s = H x over GF(q)."""
        x = np.ascontiguousarray(x, np.int32)
        s = np.zeros(self.m, np.int32)
        check(load_library().ldpc_code_syndrome(self._h, _ptr(x), _ptr(s)))
        return s

    def lds_schedule_stats(self, block, optimise=True):
        st = np.zeros(5, np.int64)
        check(load_library().ldpc_code_lds_schedule_stats(self._h, int(block), int(optimise), _ptr(st)))
        return dict(conflicts=int(st[0]), accesses=int(st[1]), words=int(st[2]), slots=int(st[3]), errors=int(st[4]))

    def vpass_place_stats(self, block):
        """Variable-pass placement of the fused floating-point min-sum decoder (host only)."""
        st = np.zeros(4, np.int64)
        check(load_library().ldpc_code_vpass_place_stats(self._h, int(block), _ptr(st)))
        return dict(before=int(st[0]), after=int(st[1]), reads=int(st[2]), errors=int(st[3]))

    def close(self):
        if self._h:
            load_library().ldpc_code_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Decoder:
    """One GPU decoder for one code and one algorithm (replaces dec())."""

    def __init__(self, code, alg, device=0, max_frames=4096, early_exit=True, want_llr=False,
                 cn_variant=0, layer_checks=0):
        L = load_library()
        self.code = code
        self.alg = alg
        o = _Opts()
        o.alg, o.device, o.max_frames = alg, device, max_frames
        o.early_exit, o.want_llr, o.cn_variant = int(early_exit), int(want_llr), cn_variant
        o.layer_checks = int(layer_checks)   # 0: flooding (the reference's schedule); L: layers of L checks
        h = C.c_void_p()
        check(L.ldpc_decoder_create(code._h, C.byref(o), C.byref(h)))
        self._h = h
        self.want_llr = want_llr
        self.max_frames = L.ldpc_decoder_max_frames(h)

    def decode(self, llr, syndrome=None, max_iter=10, want=("bits", "iters", "synd", "status")):
        """llr: [F, n] (binary) or [F, n, q]; syndrome: [F, m] uint8 or None.
        Returns a dict of numpy arrays."""
        L = load_library()
        llr = np.ascontiguousarray(llr)
        if llr.dtype not in _NP2DT:
            llr = llr.astype(np.float64)
        per = self.code.n * (self.code.q if self.code.q > 2 else 1)
        if llr.size % per:
            raise ValueError("llr size is not a multiple of the frame size")
        F = llr.size // per
        sy = None
        if syndrome is not None:
            sy = np.ascontiguousarray(syndrome, np.uint8).reshape(F, self.code.m)
        out = {}
        bits = np.zeros((F, self.code.n), np.uint8) if "bits" in want else None
        iters = np.zeros(F, np.int32) if "iters" in want else None
        synd = np.zeros((F, self.code.m), np.uint8) if "synd" in want else None
        status = np.zeros(F, np.int32) if "status" in want else None
        post = np.zeros((F, self.code.n), np.float32) if (self.want_llr and "llr" in want) else None
        check(L.ldpc_decode_batch(self._h, F, _ptr(llr), _NP2DT[llr.dtype], _ptr(sy), int(max_iter),
                                  _ptr(bits), _ptr(iters), _ptr(synd), _ptr(status), _ptr(post)))
        for k, v in (("bits", bits), ("iters", iters), ("synd", synd), ("status", status), ("llr", post)):
            if v is not None:
                out[k] = v
        return out

    def decode_device(self, nframes, llr_ptr, llr_dtype, max_iter, syndrome_ptr=None, bits_ptr=None,
                      iters_ptr=None, synd_ptr=None, status_ptr=None, llr_out_ptr=None, stream=None,
                      noalloc=False):
        """Device-pointer form (integers / c_void_p, e.g. torch.Tensor.data_ptr()).
        noalloc: the allocation-free variant (scratch in the library's static device array)."""
        vp = C.c_void_p
        L = load_library()
        fn = L.ldpc_decode_batch_device_noalloc if noalloc else L.ldpc_decode_batch_device
        check(fn(
            self._h, int(nframes), vp(llr_ptr), int(llr_dtype), vp(syndrome_ptr), int(max_iter),
            vp(bits_ptr), vp(iters_ptr), vp(synd_ptr), vp(status_ptr), vp(llr_out_ptr), vp(stream)))

    def decode_trace_q88(self, llr, syndrome, max_iter, trace_path, circ_size=128):
        L = load_library()
        llr = np.ascontiguousarray(llr)
        if llr.dtype not in _NP2DT:
            llr = llr.astype(np.float64)
        sy = None if syndrome is None else np.ascontiguousarray(syndrome, np.uint8)
        bits = np.zeros(self.code.n, np.uint8)
        synd = np.zeros(self.code.m, np.uint8)
        iters = np.zeros(1, np.int32)
        status = np.zeros(1, np.int32)
        check(L.ldpc_decode_trace_q88(self._h, _ptr(llr), _NP2DT[llr.dtype], _ptr(sy), int(max_iter),
                                      _ptr(bits), _ptr(iters), _ptr(synd), _ptr(status),
                                      os.fsencode(trace_path), int(circ_size)))
        return dict(bits=bits, synd=synd, iters=int(iters[0]), status=int(status[0]))

    def profile(self, enable=True):
        check(load_library().ldpc_decoder_profile(self._h, int(enable)))

    def profile_read(self):
        """-> (ms[4], launches[4]) per kernel class: vn, cn, fused, other."""
        ms = np.zeros(4, np.float64)
        cnt = np.zeros(4, np.int32)
        check(load_library().ldpc_decoder_profile_read(self._h, _ptr(ms), _ptr(cnt)))
        return ms, cnt

    def profile_frames(self):
        """-> frame_launches[4]: per kernel class, the frames summed over the launches of the last profile_read()."""
        fl = np.zeros(4, np.int64)
        check(load_library().ldpc_decoder_profile_frames(self._h, _ptr(fl)))
        return fl

    @property
    def last_iterations(self):
        return load_library().ldpc_decoder_last_iterations(self._h)

    def close(self):
        if self._h:
            load_library().ldpc_decoder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MonteCarlo:
    """The trial loop of the reference's *_cw_noise_gen harnesses on the device
    (include/ldpc_hip.h, ldpc_mc_*): source, enc(), channel, dec(), classification."""

    COUNTERS = ("frames", "not_converged", "undetected", "bit_errors", "iterations", "frames_in_error",
                "bit_errors_converged")
    _NP = {DT.F64: np.float64, DT.F32: np.float32, DT.I16: np.int16, DT.I32: np.int32}

    def __init__(self, code, decoder, device=0):
        L = load_library()
        self.code, self.decoder = code, decoder
        h = C.c_void_p()
        check(L.ldpc_mc_create(code._h, decoder._h, device, C.byref(h)))
        self._h = h

    @staticmethod
    def _opts(channel, param, llr_dtype, max_iter, seed, first_frame, source):
        o = _McOpts()
        o.channel, o.source, o.llr_dtype, o.max_iter = channel, source, llr_dtype, max_iter
        o.param, o.seed, o.first_frame = float(param), int(seed), int(first_frame)
        return o

    def run(self, nframes, channel, param, llr_dtype, max_iter, seed=1, first_frame=0, source=SRC.RANDOM):
        o = self._opts(channel, param, llr_dtype, max_iter, seed, first_frame, source)
        c = np.zeros(8, np.uint64)
        check(load_library().ldpc_mc_run(self._h, C.byref(o), int(nframes), _ptr(c)))
        return {k: int(c[i]) for i, k in enumerate(self.COUNTERS)}

    def generate(self, nframes, channel, param, llr_dtype, seed=1, first_frame=0, source=SRC.RANDOM):
        o = self._opts(channel, param, llr_dtype, 0, seed, first_frame, source)
        x = np.empty((nframes, self.code.n), np.uint8)
        llr = np.empty((nframes, self.code.n), self._NP[llr_dtype])
        s = np.empty((nframes, self.code.m), np.uint8)
        check(load_library().ldpc_mc_generate(self._h, C.byref(o), int(nframes), _ptr(x), _ptr(llr), _ptr(s)))
        return x, llr, s

    def set_qlc_channel(self, p_rec_given_sent, num_reads):
        """Confusion matrix [16 or 46][16], columns normalised (test_code_B_MSDP.c:455-465)."""
        p = np.ascontiguousarray(p_rec_given_sent, np.float64)
        check(load_library().ldpc_mc_set_qlc_channel(self._h, _ptr(p), int(num_reads)))

    def last_pass(self, nframes):
        x = np.empty((nframes, self.code.n), np.uint8)
        bits = np.empty((nframes, self.code.n), np.uint8)
        status = np.empty(nframes, np.int32)
        iters = np.empty(nframes, np.int32)
        check(load_library().ldpc_mc_last_pass(self._h, int(nframes), _ptr(x), _ptr(bits), _ptr(status), _ptr(iters)))
        return x, bits, status, iters

    def close(self):
        if self._h:
            load_library().ldpc_mc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GfMonteCarlo:
    """The trial loop of GFq_LDPC() on the device (include/ldpc_hip.h, ldpc_gfmc_*): source symbols, enc(),
    the confusion-matrix channel, dec(), classification."""

    COUNTERS = ("frames", "not_converged", "undetected", "symbol_errors", "iterations", "frames_in_error",
                "symbol_errors_converged")

    def __init__(self, code, decoder, device=0):
        L = load_library()
        self.code, self.decoder = code, decoder
        h = C.c_void_p()
        check(L.ldpc_gfmc_create(code._h, decoder._h, device, C.byref(h)))
        self._h = h

    def set_channel(self, p_rec_given_sent, num_reads):
        """Confusion matrix [q * num_reads][q], columns normalised (GFq_LDPC.c:548-558)."""
        p = np.ascontiguousarray(p_rec_given_sent, np.float64)
        check(load_library().ldpc_gfmc_set_channel(self._h, _ptr(p), int(num_reads)))

    def run(self, nframes, max_iter, seed=1, first_frame=0):
        c = np.zeros(8, np.uint64)
        check(load_library().ldpc_gfmc_run(self._h, int(max_iter), int(nframes), int(seed), int(first_frame), _ptr(c)))
        return {k: int(c[i]) for i, k in enumerate(self.COUNTERS)}

    def generate(self, nframes, seed=1, first_frame=0):
        n, m, q = self.code.n, self.code.m, self.code.q
        x = np.empty((nframes, n), np.uint8)
        y = np.empty((nframes, n), np.int32)
        sel = np.empty((nframes, n), np.float32)
        lf = np.empty((nframes, n, q), np.float64)
        s = np.empty((nframes, m), np.uint8)
        check(load_library().ldpc_gfmc_generate(self._h, int(seed), int(first_frame), int(nframes), _ptr(x), _ptr(y),
                                                _ptr(sel), _ptr(lf), _ptr(s)))
        return x, y, sel, lf, s

    def last_pass(self, nframes):
        x = np.empty((nframes, self.code.n), np.uint8)
        xhat = np.empty((nframes, self.code.n), np.uint8)
        status = np.empty(nframes, np.int32)
        iters = np.empty(nframes, np.int32)
        check(load_library().ldpc_gfmc_last_pass(self._h, int(nframes), _ptr(x), _ptr(xhat), _ptr(status), _ptr(iters)))
        return x, xhat, status, iters

    def close(self):
        if self._h:
            load_library().ldpc_gfmc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
