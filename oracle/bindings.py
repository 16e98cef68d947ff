"""ctypes bindings of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by ldpc_amd (the product).  Two kinds of object live here:

* ``Oracle``      -- oracle/libldpc_oracle.so, the C restatement (ldpc_oracle*.c)
* ``RefNoMexB`` / ``RefNTT`` -- oracle/_ref/*.so, the real reference programs
  test_code_no_mex_B.c / GFq_LDPC_NTT.c compiled unmodified (``-Dmain=ref_main``)
  by oracle/Makefile in the build container.  They keep their state in
  file-scope globals, so one instance per process and not re-entrant.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def build(quiet=True):
    """(Re)build the oracle; a no-op for _ref when /root/reference is absent."""
    subprocess.run(["make", "-C", HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


class _Code(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("cmax", C.c_int), ("rmax", C.c_int),
                ("q", C.c_int),
                ("col_weight", C.POINTER(C.c_int)), ("row_weight", C.POINTER(C.c_int)),
                ("row_col", C.POINTER(C.c_int)), ("row_N", C.POINTER(C.c_int)),
                ("col_row", C.POINTER(C.c_int)), ("col_N", C.POINTER(C.c_int)),
                ("H", C.POINTER(C.c_int))]


class OracleCode:
    def __init__(self, lib, path, dialect=0):
        self._lib = lib
        p = C.POINTER(_Code)()
        rc = lib.ora_load_alist(os.fsencode(path), dialect, C.byref(p))
        if rc != 0:
            raise OSError(f"oracle: cannot load alist {path!r} (rc={rc})")
        self._p = p
        c = p.contents
        self.n, self.m, self.cmax, self.rmax, self.q = c.n, c.m, c.cmax, c.rmax, c.q
        self.col_weight = np.ctypeslib.as_array(c.col_weight, (self.n,)).copy()
        self.row_weight = np.ctypeslib.as_array(c.row_weight, (self.m,)).copy()
        self.row_col = np.ctypeslib.as_array(c.row_col, (self.m, self.rmax)).copy()
        self.row_N = np.ctypeslib.as_array(c.row_N, (self.m, self.rmax)).copy()
        self.col_row = np.ctypeslib.as_array(c.col_row, (self.n, self.cmax)).copy()
        self.col_N = np.ctypeslib.as_array(c.col_N, (self.n, self.cmax)).copy()
        self.H = np.ctypeslib.as_array(c.H, (self.m, self.rmax)).copy()

    def __del__(self):
        if getattr(self, "_p", None):
            self._lib.ora_free_code(self._p)
            self._p = None


class Oracle:
    def __init__(self):
        path = os.path.join(HERE, "libldpc_oracle.so")
        if not os.path.exists(path):
            build()
        L = self.lib = C.CDLL(path)
        cp = C.POINTER(_Code)
        L.ora_load_alist.argtypes = [C.c_char_p, C.c_int, C.POINTER(cp)]
        L.ora_free_code.argtypes = [cp]
        L.ora_enc.argtypes = [cp, _i32p, _i32p]
        L.ora_float_to_fix.argtypes = [C.c_double]
        L.ora_float_to_fix.restype = C.c_double
        ip = C.POINTER(C.c_int)
        L.ora_dec_minsum.argtypes = [cp, _f64p, _i32p, C.c_int, _i32p, _i32p, ip, C.c_void_p]
        L.ora_dec_minsum_f32.argtypes = [cp, _f32p, _i32p, C.c_int, _i32p, _i32p, ip, C.c_void_p]
        L.ora_dec_minsum_fixed.argtypes = [cp, _f64p, _i32p, C.c_int, _i32p, _i32p, ip,
                                           C.c_void_p, C.c_void_p, C.c_int]
        L.ora_dec_minsum_fixed_layered.argtypes = [cp, _f64p, _i32p, C.c_int, C.c_int, _i32p, _i32p, ip, C.c_void_p]
        L.ora_dec_spa.argtypes = [cp, _f64p, _i32p, C.c_int, _i32p, _i32p, ip, C.c_void_p]
        L.ora_dec_spa_f32.argtypes = [cp, _f32p, _i32p, C.c_int, _i32p, _i32p, ip, C.c_void_p]
        L.ora_dec_spa_table.argtypes = [cp, _i32p, _i32p, C.c_int, _i32p, _i32p, ip, C.c_void_p]
        L.ora_inittab.argtypes = [C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int))]
        L.ora_float2fix14.argtypes = [C.c_double]
        L.ora_gf_tables.argtypes = [C.c_int, _i32p, _i32p]
        L.ora_gf_mul.argtypes = [C.c_int, C.c_int, C.c_int]
        L.ora_gf_enc.argtypes = [cp, C.c_int, _i32p, _i32p]
        L.ora_wht_tables.argtypes = [C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_uint))]
        L.ora_wht_float2fix.argtypes = [C.c_double]
        L.ora_dec_gf2p_wht.argtypes = [cp, C.c_int, _i32p, _i32p, C.c_int, _i32p, _i32p, ip]
        L.ora_dec_gfq_direct.argtypes = [cp, C.c_int, _f64p, _i32p, C.c_int, _i32p, _i32p, ip,
                                         C.c_void_p]
        L.ora_gf_normalise_confusion.argtypes = [C.c_int, C.c_int, _f64p]
        L.ora_gf_assign_llr.argtypes = [C.c_int, C.c_int, _f64p, _i32p, C.c_int, _f64p]
        self._libc = C.CDLL(None)
        self._libc.fopen.restype = C.c_void_p
        self._libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        self._libc.fclose.argtypes = [C.c_void_p]

    def load(self, path, dialect=0):
        return OracleCode(self.lib, path, dialect)

    def enc(self, code, y):
        s = np.zeros(code.m, np.int32)
        self.lib.ora_enc(code._p, np.ascontiguousarray(y, np.int32), s)
        return s

    def float_to_fix(self, v):
        return np.array([self.lib.ora_float_to_fix(float(x)) for x in np.ravel(v)]).reshape(np.shape(v))

    # --- binary decoders: one frame; returns dict(rc, bits, synd, iters, post)
    def _run(self, fn, code, q0, s, loop_max, qdt, pdt, extra=()):
        q0 = np.ascontiguousarray(q0, qdt)
        s = np.zeros(code.m, np.int32) if s is None else np.ascontiguousarray(s, np.int32)
        bits = np.zeros(code.n, np.int32)
        synd = np.zeros(code.m, np.int32)
        it = C.c_int(0)
        post = np.zeros(code.n, pdt)
        rc = fn(code._p, q0, s, int(loop_max), bits, synd, C.byref(it),
                post.ctypes.data_as(C.c_void_p), *extra)
        return dict(rc=rc, bits=bits, synd=synd, iters=it.value, post=post)

    def dec_minsum(self, code, q0, s, loop_max):
        return self._run(self.lib.ora_dec_minsum, code, q0, s, loop_max, np.float64, np.float64)

    def dec_minsum_f32(self, code, q0, s, loop_max):
        return self._run(self.lib.ora_dec_minsum_f32, code, q0, s, loop_max, np.float32, np.float32)

    def dec_minsum_fixed(self, code, q0, s, loop_max, trace_path=None, circ_size=128):
        fp = None
        if trace_path is not None:
            fp = self._libc.fopen(os.fsencode(trace_path), b"wt")
            if not fp:
                raise OSError(f"cannot open {trace_path}")
        try:
            return self._run(self.lib.ora_dec_minsum_fixed, code, q0, s, loop_max,
                             np.float64, np.float64, (C.c_void_p(fp), int(circ_size)))
        finally:
            if fp:
                self._libc.fclose(fp)

    def dec_minsum_fixed_layered(self, code, q0, s, loop_max, layer):
        """Q8.8 min-sum on a layered schedule (layers of `layer` consecutive checks)."""
        q0 = np.ascontiguousarray(q0, np.float64)
        s = np.zeros(code.m, np.int32) if s is None else np.ascontiguousarray(s, np.int32)
        bits = np.zeros(code.n, np.int32)
        synd = np.zeros(code.m, np.int32)
        it = C.c_int(0)
        post = np.zeros(code.n, np.float64)
        rc = self.lib.ora_dec_minsum_fixed_layered(code._p, q0, s, int(loop_max), int(layer), bits, synd,
                                                   C.byref(it), post.ctypes.data_as(C.c_void_p))
        return dict(rc=rc, bits=bits, synd=synd, iters=it.value, post=post)

    def dec_spa(self, code, q0, s, loop_max):
        return self._run(self.lib.ora_dec_spa, code, q0, s, loop_max, np.float64, np.float64)

    def dec_spa_f32(self, code, q0, s, loop_max):
        return self._run(self.lib.ora_dec_spa_f32, code, q0, s, loop_max, np.float32, np.float32)

    def dec_spa_table(self, code, q0, s, loop_max):
        return self._run(self.lib.ora_dec_spa_table, code, q0, s, loop_max, np.int32, np.int32)

    def inittab(self):
        a = C.POINTER(C.c_int)()
        b = C.POINTER(C.c_int)()
        self.lib.ora_inittab(C.byref(a), C.byref(b))
        return (np.ctypeslib.as_array(a, (1 << 22,)), np.ctypeslib.as_array(b, (1 << 22,)))

    def float2fix14(self, x):
        return np.array([self.lib.ora_float2fix14(float(v)) for v in np.ravel(x)],
                        np.int32).reshape(np.shape(x))

    # --- non-binary
    def gf_tables(self, q):
        lg = np.zeros(q, np.int32)
        ex = np.zeros(q, np.int32)
        self.lib.ora_gf_tables(q, lg, ex)
        return lg, ex[:q - 1]

    def gf_mul(self, q, a, b):
        return self.lib.ora_gf_mul(q, a, b)

    def gf_enc(self, code, q, x):
        s = np.zeros(code.m, np.int32)
        self.lib.ora_gf_enc(code._p, q, np.ascontiguousarray(x, np.int32), s)
        return s

    def wht_tables(self):
        a = C.POINTER(C.c_int)()
        b = C.POINTER(C.c_uint)()
        self.lib.ora_wht_tables(C.byref(a), C.byref(b))
        flog = np.ctypeslib.as_array(a, (1 << 20,))
        base = C.cast(C.addressof(b.contents) - 4 * (1 << 20), C.POINTER(C.c_uint))
        fexp = np.ctypeslib.as_array(base, (1 << 21,))
        return flog, fexp          # fexp[i + 2^20] = Fexp(i)

    def wht_float2fix(self, x):
        return np.array([self.lib.ora_wht_float2fix(float(v)) for v in np.ravel(x)],
                        np.int32).reshape(np.shape(x))

    def dec_gf2p_wht(self, code, log2q, logfna, z, loop_max):
        logfna = np.ascontiguousarray(logfna, np.int32)
        z = np.ascontiguousarray(z, np.int32)
        xh = np.zeros(code.n, np.int32)
        zh = np.zeros(code.m, np.int32)
        it = C.c_int(0)
        rc = self.lib.ora_dec_gf2p_wht(code._p, log2q, logfna, z, int(loop_max), xh, zh, C.byref(it))
        return dict(rc=rc, x=xh, z=zh, iters=it.value)

    def dec_gfq_direct(self, code, q, logfna, z, loop_max):
        logfna = np.ascontiguousarray(logfna, np.float64)      # [q][n]
        z = np.ascontiguousarray(z, np.int32)
        xh = np.zeros(code.n, np.int32)
        zh = np.zeros(code.m, np.int32)
        it = C.c_int(0)
        post = np.zeros((q, code.n), np.float64)
        rc = self.lib.ora_dec_gfq_direct(code._p, q, logfna, z, int(loop_max), xh, zh,
                                         C.byref(it), post.ctypes.data_as(C.c_void_p))
        return dict(rc=rc, x=xh, z=zh, iters=it.value, post=post)

    def gf_channel_draw(self, q, num_reads, conf_normalised, x, rand_select):
        """channel() :213-231 -> y; conf_normalised as ora_gf_normalise_confusion leaves it."""
        conf = np.ascontiguousarray(conf_normalised, np.float64)
        x = np.ascontiguousarray(x, np.int32)
        rs = np.ascontiguousarray(rand_select, np.float32)
        y = np.zeros(len(x), np.int32)
        self.lib.ora_gf_channel_draw.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self.lib.ora_gf_channel_draw(q, num_reads, conf.ctypes.data, x.ctypes.data, rs.ctypes.data, len(x), y.ctypes.data)
        return y

    def gf_assign_llr(self, q, num_reads, conf, y):
        conf = np.array(conf, np.float64, order="C").copy()
        self.lib.ora_gf_normalise_confusion(q, num_reads, conf)
        out = np.zeros((q, len(y)), np.float64)
        self.lib.ora_gf_assign_llr(q, num_reads, conf, np.ascontiguousarray(y, np.int32), len(y), out)
        return out


# --------------------------------------------------------------- oracle/_ref

def _ref_path(name):
    return os.path.join(HERE, "_ref", name)


def have_ref(name):
    return os.path.exists(_ref_path(name))


class RefNoMexB:
    """The reference's test_code_no_mex_B.c (table sum-product), built unmodified."""

    def __init__(self):
        L = self.lib = C.CDLL(_ref_path("libref_no_mex_B.so"))
        L.initdec.argtypes = [C.c_char_p]
        L.dec.argtypes = [_i32p, _i32p, C.c_int, _i32p]
        L.float2fix.argtypes = [C.c_double]
        L.inittab()
        self._loaded = None

    def load(self, path):
        self.lib.initdec(os.fsencode(path))      # leaks the previous graph, like the reference
        self.n = C.c_int.in_dll(self.lib, "n").value
        self.m = C.c_int.in_dll(self.lib, "m").value

    def float2fix(self, x):
        return self.lib.float2fix(float(x))

    def dec(self, q0, s, loop_max):
        q0 = np.ascontiguousarray(q0, np.int32)
        s = np.ascontiguousarray(s, np.int32)
        x = np.zeros(self.n, np.int32)
        rc = self.lib.dec(q0, s, int(loop_max), x)
        tb = C.POINTER(C.c_int).in_dll(self.lib, "tmp_bit")
        ts = C.POINTER(C.c_int).in_dll(self.lib, "tmp_s")
        loop = C.c_int.in_dll(self.lib, "loop").value
        return dict(rc=rc, bits=np.ctypeslib.as_array(tb, (self.n,)).copy(),
                    synd=np.ctypeslib.as_array(ts, (self.m,)).copy(),
                    iters=loop + 1 if rc == 0 else loop)


class RefNTT:
    """The reference's GFq_LDPC_NTT.c built with -DLog2Q=p (p = 2 or 3)."""

    def __init__(self, log2q):
        self.log2q = log2q
        self.Q = 1 << log2q
        L = self.lib = C.CDLL(_ref_path(f"libref_ntt_q{self.Q}.so"))
        L.initdec.argtypes = [C.c_char_p]
        L.SRand.argtypes = [C.c_int]
        L.Rand.restype = C.c_uint
        L.initlogexptab2()

    def load(self, path):
        self.lib.initdec(os.fsencode(path))
        self.n = C.c_int.in_dll(self.lib, "n").value
        self.m = C.c_int.in_dll(self.lib, "m").value

    def _rows(self, arr2d):
        rows = (C.POINTER(C.c_int) * arr2d.shape[0])()
        for i in range(arr2d.shape[0]):
            rows[i] = arr2d[i].ctypes.data_as(C.POINTER(C.c_int))
        return rows

    def experiment(self, seed, channel, level):
        """main():681-695 for one seed: returns x, s, logfna [n][Q]."""
        L = self.lib
        L.SRand(seed)
        x = np.array([L.Rand() % self.Q for _ in range(self.n)], np.int32)
        s = np.zeros(self.m, np.int32)
        L.enc(x.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p))
        y = np.zeros(self.n, np.int32)
        logfna = np.zeros((self.n, self.Q), np.int32)
        rows = self._rows(logfna)
        fn = getattr(L, channel)
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        fn(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), float(level), rows)
        return x, s, y, logfna

    def dec(self, logfna, z, loop_max):
        logfna = np.ascontiguousarray(logfna, np.int32)
        rows = self._rows(logfna)
        z = np.ascontiguousarray(z, np.int32)
        x = np.zeros(self.n, np.int32)
        self.lib.dec.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        rc = self.lib.dec(rows, z.ctypes.data_as(C.c_void_p), int(loop_max),
                          x.ctypes.data_as(C.c_void_p))
        tx = C.POINTER(C.c_int).in_dll(self.lib, "tmp_x")
        tz = C.POINTER(C.c_int).in_dll(self.lib, "tmp_z")
        return dict(rc=rc, x=np.ctypeslib.as_array(tx, (self.n,)).copy(),
                    z=np.ctypeslib.as_array(tz, (self.m,)).copy())
