"""The reference's Python decoder API, served by the GPU library.

``run(H, LLR_init, max_iterations)`` has the signature and return shape of
LDPC_decode.py:348-374 (and parallel.py:291-314): sum-product decoding of one
word against the zero syndrome, returning ``(cw, iterations, seconds)`` with
``cw`` an (n, 1) int array, ``iterations`` the 1-based iteration at which
H.cw = 0 was reached or ``False`` when max_iterations ran out.  ``H`` may be a
dense matrix (as there), an alist path or an ldpc_amd.Code.
"""
import time

import numpy as np

from .lib import ALG, Code, Decoder

_cache = {}


def _decoder_for(H, alg, max_frames):
    if isinstance(H, Code):
        key = (id(H), alg, max_frames)
        code = H
    elif isinstance(H, (str, bytes)):
        key = (H, alg, max_frames)
        code = None
    else:
        Hd = np.asarray(H)
        key = (Hd.shape, Hd.tobytes(), alg, max_frames)
        code = None
    if key not in _cache:
        if code is None:
            code = Code.from_alist(H) if isinstance(H, (str, bytes)) else Code.from_dense(H)
        _cache[key] = Decoder(code, alg, max_frames=max_frames, early_exit=True)
    return _cache[key]


def LLR_to_cw(LLRs):
    """LDPC_decode.py:282-292: (n,1) hard decisions, LLR == 0 -> 0."""
    return (np.asarray(LLRs, dtype=float).reshape(-1, 1) < 0).astype(int)


def run(H, LLR_init, max_iterations, precheck=False, alg=ALG.SPA_F64):
    """Decode one word.  precheck=True adds parallel.py:293-295's shortcut: when
    the hard decision of LLR_init already satisfies H the iteration slot is True."""
    t1 = time.perf_counter()
    dec = _decoder_for(H, alg, 64)
    llr = np.ascontiguousarray(LLR_init, np.float64).reshape(1, -1)
    if precheck:
        cw0 = (llr[0] < 0).astype(np.int32)
        if not dec.code.syndrome(cw0).any():
            return (cw0.reshape(-1, 1).astype(int), True, time.perf_counter() - t1)
    out = dec.decode(llr, None, max_iterations)
    cw = out["bits"][0].astype(int).reshape(-1, 1)
    it = int(out["iters"][0]) if out["status"][0] == 0 else False
    return (cw, it, time.perf_counter() - t1)


def decode(H, LLRs, max_num_iterations):
    """LDPC_decode.py:37-66 (the scipy-sparse variant of run): same result shape."""
    return run(H, LLRs, max_num_iterations)


def run_batch(H, LLRs, max_iterations, syndrome=None, alg=ALG.MINSUM_F32, max_frames=4096):
    """Batched form: LLRs [F, n] -> dict(bits, iters, synd, status)."""
    dec = _decoder_for(H, alg, max_frames)
    return dec.decode(LLRs, syndrome, max_iterations)
