"""Is the fused Q8.8 kernel paced by its busiest SIMD?  Random (n=16000, dv=3) codes with 24, 25
and 28 wave-checks (m = 1536, 1600, 1792; all take the 2-checks-per-thread, 32-slot kernel):
per-iteration time should follow max SIMD load 6, 7, 7 if so."""
import os, sys, time, math
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ldpc_amd
from ldpc_amd import ALG, DT
from test_gpu_binary import _random_code
dev = torch.device("cuda", 0)
F = 32768
for m in (1536, 1600, 1792, 2048):
    rng = np.random.default_rng(m)
    H = _random_code(rng, 16000, m, 3)
    path = f"/tmp/rand_{m}.alist"
    ldpc_amd.alist.matrix_to_alist(H, path)
    code = ldpc_amd.Code.from_alist(path)
    dec = ldpc_amd.Decoder(code, ALG.MINSUM_Q88, max_frames=F, early_exit=False)
    g = torch.Generator(device=dev); g.manual_seed(1)
    Lraw = int(round(math.log(0.98 / 0.02) * 256))
    llr = torch.where(torch.rand((F, code.n), device=dev, generator=g) < 0.02, -Lraw, Lraw).to(torch.int16).contiguous()
    bits = torch.empty((F, code.n), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for it in (2, 10):
        dec.decode_device(F, llr.data_ptr(), DT.I16, it, None, bits.data_ptr(), None, None, None, None, st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4):
            dec.decode_device(F, llr.data_ptr(), DT.I16, it, None, bits.data_ptr(), None, None, None, None, st)
        torch.cuda.synchronize(); res[it] = (time.perf_counter() - t0) / 4
    per_it = (res[10] - res[2]) / 8
    print(f"m={m} rmax={code.rmax} wave-checks={math.ceil(m/64)}: {per_it*1e3:.3f} ms per iteration per {F} frames, {per_it/F*256*1e6:.2f} us per frame-iteration per CU", flush=True)
