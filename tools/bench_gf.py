#!/usr/bin/env python3
"""Throughput of the non-binary engines (GF2P_WHT integer Walsh-Hadamard, GFQ_DIRECT
log-domain) with frames resident in HBM; 10 iterations executed (noisy priors, no early exit).
  python tools/bench_gf.py > gpurun_out/bench_gf.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CODES = os.path.join(ROOT, "tests", "golden", "codes")


def main():
    import torch
    import ldpc_amd
    from ldpc_amd import ALG, DT
    dev = torch.device("cuda", 0)
    cases = [("q4.sp.9000.6000.4500.1", "GF2P_WHT", DT.I32, torch.int32, 2048),
             ("q8.sp.6000.4000.3000.1", "GF2P_WHT", DT.I32, torch.int32, 2048),
             ("q4.sp.9000.6000.4500.1", "GFQ_DIRECT", DT.F32, torch.float32, 2048),
             ("peg_4376_4_0.9_NB.txt", "GFQ_DIRECT", DT.F32, torch.float32, 1024)]
    print(f"{'code':26s} {'alg':11s} {'q':>3s} {'frames':>6s} {'ms/pass':>9s} {'Msym/s':>9s} {'info Gbit/s':>11s}")
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, alg, dt, tdt, F in cases:
        if want not in name + alg:
            continue
        F = int(os.environ.get("BENCH_GF_FRAMES", F))      # (frames per call: sweeps)
        code = ldpc_amd.Code.from_alist(os.path.join(CODES, name))
        n, m, q = code.n, code.m, code.q
        gen = torch.Generator(device=dev)
        gen.manual_seed(3)
        u = torch.rand((F, n, q), device=dev, generator=gen)
        if alg == "GF2P_WHT":
            llr = (torch.log2(u / u.sum(dim=2, keepdim=True)) * 16384).to(tdt).contiguous()     # Q.14 log2 priors
        else:
            llr = torch.log(u / u.sum(dim=2, keepdim=True)).to(tdt).contiguous()
        bits = torch.empty((F, n), dtype=torch.uint8, device=dev)
        status = torch.empty(F, dtype=torch.int32, device=dev)
        dec = ldpc_amd.Decoder(code, getattr(ALG, alg), max_frames=F, early_exit=False)
        st = torch.cuda.current_stream().cuda_stream

        def step():
            dec.decode_device(F, llr.data_ptr(), dt, 10, None, bits.data_ptr(), None, None, status.data_ptr(), None, st)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / 3
        dec.profile(True)
        step()
        ms_k, cnt_k = dec.profile_read()
        dec.profile(False)
        print(f"{name:26s} {alg:11s} {q:3d} {F:6d} {dt_s * 1e3:9.3f} {n * F / dt_s / 1e6:9.1f} "
              f"{(n - m) * np.log2(q) * F / dt_s / 1e9:11.3f}   per pass: check {ms_k[1]:.1f} ms, variable {ms_k[0]:.1f} ms",
              flush=True)


if __name__ == "__main__":
    main()
