#!/usr/bin/env python3
"""Throughput of every decoder variant on the repository's codes (not the
headline benchmark -- that is bench.py).  Frames resident in HBM, 10 iterations, no early
exit, BSC-like LLRs above threshold so that every frame runs all iterations.

  python tools/bench_codes.py [code-substring [alg-substring]] > gpurun_out/bench_codes.txt
"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CODES = os.path.join(ROOT, "tests", "golden", "codes")


ITERS = int(os.environ.get("BENCH_CODES_ITERS", "10"))   # iterations executed per frame


def main():
    import torch
    import ldpc_amd
    from ldpc_amd import ALG, DT
    dev = torch.device("cuda", 0)
    rows = [("peg_16000_3_0.9.txt", 0.02), ("32000.2240.3.105.txt", 0.02), ("4095.737.3.txt", 0.03),
            ("1057.244.3.txt", 0.04), ("Frolov_1024_0.5.txt", 0.12), ("204.33.484.txt", 0.12), ("dvb_H_0.5.txt", 0.12)]
    algs = [("MINSUM_Q88", DT.I16, torch.int16), ("MINSUM_F32", DT.F32, torch.float32),
            ("MINSUM_F64", DT.F64, torch.float64), ("SPA_F32", DT.F32, torch.float32), ("SPA_F64", DT.F64, torch.float64),
            ("SPA_TABLE_Q14", DT.I32, torch.int32)]
    want_code = sys.argv[1] if len(sys.argv) > 1 else ""
    want_alg = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = [r for r in rows if want_code in r[0]]
    algs = [a for a in algs if want_alg in a[0]]
    print(f"{'code':24s} {'alg':14s} {'frames':>7s} {'ms/pass':>9s} {'info Gbit/s':>12s}")
    for name, p in rows:
        code = ldpc_amd.Code.from_alist(os.path.join(CODES, name))
        n, m = code.n, code.m
        L = math.log((1 - p) / p)
        for alg, dt, tdt in algs:
            budget = float(os.environ.get("BENCH_CODES_BUDGET", 2.0e9 if alg.startswith("MINSUM") else 1.2e9))        # bytes of edge messages resident
            F = int(min(65536, max(256, budget // (code.E * 8))))
            F -= F % 64
            if os.environ.get("BENCH_CODES_FRAMES"):
                F = int(os.environ["BENCH_CODES_FRAMES"])
            gen = torch.Generator(device=dev)
            gen.manual_seed(1)
            flips = torch.rand((F, n), device=dev, generator=gen) < p
            if dt == DT.I16:
                llr = torch.where(flips, -int(round(L * 256)), int(round(L * 256))).to(tdt)
            elif dt == DT.I32:
                llr = torch.where(flips, -int(L * 16384 + 0.5), int(L * 16384 + 0.5)).to(tdt)
            else:
                llr = torch.where(flips, -L, L).to(tdt)
                if os.environ.get("BENCH_CODES_LLR", "bsc") == "awgn":
                    # continuous magnitudes (BPSK + Gaussian noise at a hopeless SNR, so that every frame
                    # still runs all iterations): lanes of a wavefront then differ in the value-dependent
                    # branches of the sum-product functions, which equal-magnitude BSC inputs never show
                    sg = 1.6
                    llr = ((1.0 + sg * torch.randn((F, n), device=dev, generator=gen)) * (2.0 / sg ** 2)).to(tdt)
            llr = llr.contiguous()
            bits = torch.empty((F, n), dtype=torch.uint8, device=dev)
            status = torch.empty(F, dtype=torch.int32, device=dev)
            try:
                dec = ldpc_amd.Decoder(code, getattr(ALG, alg), max_frames=F, early_exit=False)
            except ldpc_amd.LdpcError as e:
                print(f"{name:24s} {alg:14s} unsupported: {e}")
                continue
            st = torch.cuda.current_stream().cuda_stream

            def step():
                dec.decode_device(F, llr.data_ptr(), dt, ITERS, None, bits.data_ptr(), None, None, status.data_ptr(), None, st)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                step()
            torch.cuda.synchronize()
            dt_s = (time.perf_counter() - t0) / reps
            print(f"{name:24s} {alg:14s} {F:7d} {dt_s * 1e3:9.3f} {(n - m) * F / dt_s / 1e9:12.3f}", flush=True)
            del dec, llr, bits


if __name__ == "__main__":
    main()
