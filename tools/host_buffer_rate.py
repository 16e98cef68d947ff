#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (ldpc_decode_batch): PEG-16000 Q8.8,
10 iterations executed, LLRs in pageable host memory, bits / iterations / status copied back.
Never bench.py's `value` (that one has the inputs resident in HBM); recorded in DESIGN.md."""
import math
import os
import sys
import time

import numpy as np
import torch  # before the library: one HIP runtime per process, the one torch brings (as in bench.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ldpc_amd  # noqa: E402
from ldpc_amd import ALG  # noqa: E402

torch.cuda.init()
code = ldpc_amd.Code.from_alist(os.path.join(ROOT, "tests", "golden", "codes", "peg_16000_3_0.9.txt"))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
p = 0.02
Lraw = int(round(math.log((1 - p) / p) * 256))
rng = np.random.default_rng(1)
llr = np.where(rng.random((F, code.n)) < p, -Lraw, Lraw).astype(np.int16)
dec = ldpc_amd.Decoder(code, ALG.MINSUM_Q88, max_frames=F, early_exit=False)
dec.decode(llr, None, 10, want=("bits", "iters", "status"))
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    out = dec.decode(llr, None, 10, want=("bits", "iters", "status"))
dt = (time.perf_counter() - t0) / reps
print(f"host-buffer ldpc_decode_batch: {F} frames, {dt * 1e3:.1f} ms per call, "
      f"{(code.n - code.m) * F / dt / 1e9:.2f} Gbit/s information (PCIe and host copies included); "
      f"avg iterations {out['iters'].mean():.1f}")

# the same call on pinned host memory (what a caller that cares would hand over)
import ctypes as C  # noqa: E402
L = ldpc_amd.load_library()
pl = torch.empty((F, code.n), dtype=torch.int16).pin_memory()
pl.copy_(torch.from_numpy(llr))
pb = torch.empty((F, code.n), dtype=torch.uint8).pin_memory()
pi = torch.empty(F, dtype=torch.int32).pin_memory()
ps = torch.empty(F, dtype=torch.int32).pin_memory()
vp = C.c_void_p


def call():
    rc = L.ldpc_decode_batch(dec._h, F, vp(pl.data_ptr()), ldpc_amd.DT.I16, None, 10, vp(pb.data_ptr()),
                             vp(pi.data_ptr()), None, vp(ps.data_ptr()), None)
    assert rc == 0, L.ldpc_last_error()


call()
t0 = time.perf_counter()
for _ in range(reps):
    call()
dt = (time.perf_counter() - t0) / reps
assert np.array_equal(pb.numpy(), out["bits"]) and np.array_equal(pi.numpy(), out["iters"])
print(f"same call on pinned host buffers: {dt * 1e3:.1f} ms per call, {(code.n - code.m) * F / dt / 1e9:.2f} Gbit/s information")
