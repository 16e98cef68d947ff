#!/usr/bin/env python3
"""FER sweep over a channel parameter on the device Monte-Carlo loop, frames sharded over
the ranks -- what test_mex_B_float.m:9-36 / test_mex_B.m / parallel.py:385-396 do with
parfor / Pool around the mex harness.  One JSON line per point (checkpoint + resume).

  python tools/sweep.py --code tests/golden/codes/204.33.484.txt --alg MINSUM_F32 \
      --channel awgn --ebn0 1.0:3.0:0.5 --iters 50 --frames 200000 --out sweep.jsonl
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
      tools/sweep.py --code tests/golden/codes/dvb_H_0.5.txt --alg SPA_F32 --channel awgn \
      --ebn0 0.6:1.4:0.2 --iters 50 --frames 100000 --min-failures 100 --out dvb.jsonl
  python tools/sweep.py --code .../Frolov_1024_0.5.txt --alg MINSUM_Q88 --channel bsc \
      --p 0.08:0.04:-0.0025 --iters 30 --frames 100000      # test_mex_B_float.m's own grid
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def grid(spec):
    """'a:b:step' (b inclusive, either direction) or a comma list."""
    if ":" in spec:
        a, b, s = (float(v) for v in spec.split(":"))
        if s == 0 or (b - a) * s < 0:
            raise SystemExit(f"bad grid {spec}")
        k = int(round((b - a) / s))
        return [round(a + i * s, 12) for i in range(k + 1)]
    return [float(v) for v in spec.split(",")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--code", required=True)
    ap.add_argument("--alg", default="MINSUM_F32")
    ap.add_argument("--channel", default="bsc", choices=["bsc", "bsc_iid", "awgn"])
    ap.add_argument("--p", help="BSC crossover grid")
    ap.add_argument("--ebn0", help="AWGN Eb/N0 grid in dB (rate = (n-m)/n)")
    ap.add_argument("--sigma", help="AWGN noise standard deviation grid")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--frames", type=int, default=100000, help="trials per point (all ranks together)")
    ap.add_argument("--min-failures", type=int, default=0)
    ap.add_argument("--batch", type=int, default=4096, help="frames per device pass")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-resume", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("tools/sweep.py needs a GPU: libldpc_hip has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    import ldpc_amd
    from ldpc_amd import ALG, CH, DT
    from ldpc_amd.sweep import ebn0_to_sigma, sweep

    code = ldpc_amd.Code.from_alist(args.code)
    alg = getattr(ALG, args.alg)
    dtype = {ALG.MINSUM_Q88: DT.I16, ALG.SPA_TABLE_Q14: DT.I32, ALG.MINSUM_F64: DT.F64,
             ALG.SPA_F64: DT.F64}.get(alg, DT.F32)
    dec = ldpc_amd.Decoder(code, alg, device=local_rank, max_frames=args.batch, early_exit=True)
    mc = ldpc_amd.MonteCarlo(code, dec, device=local_rank)
    if args.channel == "awgn":
        if args.ebn0:
            params = grid(args.ebn0)
            rate = (code.n - code.m) / code.n
            mcp, label = [ebn0_to_sigma(e, rate) for e in params], "Eb/N0 dB"
        else:
            params = grid(args.sigma or "0.8")
            mcp, label = params, "sigma"
        ch = CH.AWGN
    else:
        params = grid(args.p or "0.05")
        mcp, label = params, "p"
        ch = CH.BSC_EXACT if args.channel == "bsc" else CH.BSC_IID
    recs = sweep(mc, params, args.frames, ch, dtype, args.iters, seed=args.seed, out_path=args.out,
                 resume=not args.no_resume, min_failures=args.min_failures or None, device=dev,
                 label=label, mc_params=mcp)
    if rank == 0:
        for r in recs:
            print(json.dumps({k: r.get(k) for k in ("param", "frames", "failures", "undetected", "fer",
                                                    "ber", "avg_iterations", "seconds")}), flush=True)
    mc.close()
    dec.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
