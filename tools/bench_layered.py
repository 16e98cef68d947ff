#!/usr/bin/env python3
"""Flooding against layered Q8.8 min-sum (ldpc_opts.layer_checks) on one code at one BSC point:
frame failures, mean iterations and wall time per batch, inputs resident in HBM.

  python tools/bench_layered.py [code] [p] [max_iter] [frames] [layers...]
"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import ldpc_amd
    from ldpc_amd import ALG, DT
    name = sys.argv[1] if len(sys.argv) > 1 else "peg_16000_3_0.9.txt"
    p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.004
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    F = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
    layers = [int(x) for x in sys.argv[5:]] or [0, 1024, 512, 256]
    code = ldpc_amd.Code.from_alist(os.path.join(ROOT, "tests", "golden", "codes", name))
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    L = int(round(math.log((1 - p) / p) * 256))
    flips = torch.rand((F, code.n), device=dev, generator=gen) < p
    llr = torch.where(flips, -L, L).to(torch.int16).contiguous()
    bits = torch.empty((F, code.n), dtype=torch.uint8, device=dev)
    status = torch.empty(F, dtype=torch.int32, device=dev)
    its = torch.empty(F, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    print(f"{name}: BSC p={p}, {F} frames, at most {iters} iterations")
    print(f"{'layer_checks':>12s} {'failures':>9s} {'bit errors':>11s} {'mean iters':>11s} {'ms/batch':>9s} {'info Gbit/s':>12s}")
    for lc in layers:
        dec = ldpc_amd.Decoder(code, ALG.MINSUM_Q88, max_frames=F, early_exit=True, layer_checks=lc)

        def step():
            dec.decode_device(F, llr.data_ptr(), DT.I16, iters, None, bits.data_ptr(), its.data_ptr(), None, status.data_ptr(), None, st)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{lc:12d} {int((status != 0).sum()):9d} {int(bits.sum()):11d} {float(its.float().mean()):11.2f} {dt * 1e3:9.2f} "
              f"{(code.n - code.m) * F / dt / 1e9:12.2f}", flush=True)


if __name__ == "__main__":
    main()
