#!/usr/bin/env python3
"""bench.py -- headline benchmark of libldpc_hip (BASELINE.json's metric).

Workload (configs[1]): PEG peg_16000_3_0.9 (n=16000, m=1600, K=14400), fixed
point (Q8.8) min-sum -- the reference's test_code_min_sum_B_fixed.c dec() -- with
exactly 10 iterations executed per frame: all-zero codeword, zero syndrome,
BSC p=0.02 (above threshold, so no frame converges and early exit never
fires), LLR = +-log((1-p)/p) on the Q8.8 grid.  A "step" decodes one batch of
--frames frames per GPU; inputs (raw int16 LLRs, frame-major as the callers of
dec() hold them) are resident in HBM before the timed region; outputs (decoded
bits, iteration counts, status) are written to HBM.

  python bench.py [--gpus N --steps K --warmup W --frames F --alg q88|f32|f64 --p 0.02]

--gpus N > 1 without a launcher's environment: this process starts N ranks of itself
(`python -m torch.distributed.run --nproc-per-node N`, one rank per GPU, RCCL) before
anything touches the GPU, and passes rank 0's line through.  Frames shard over the ranks
with no data-path exchange (the reference's parfor batches, test_mex_B_float.m:24-30); the
only collectives are the sum of the error counters (its `sum` over the workers' results,
:73-78) and the max of the elapsed time.

Prints ONE JSON line on rank 0.  value = decoded information Gbit/s over all
ranks = K * frames * steps * ranks / seconds.

"roofline" describes the dominant kernel of the timed region (its mean launch time from
hipEvents on the launch stream):
  * fused frame-resident kernels keep every message in LDS / VGPRs, so HBM is not what
    bounds them: bound = "valu", achieved = vector wave-instructions per second
    (SQ_INSTS_VALU per launch, counted by a rocprofv3 --pmc pass of this same command,
    over the launch time), peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction
    (MI355X_MICROARCH.md).  The SURVEY 8(d) figure -- algorithmic bytes of two-pass
    flooding over the launch time -- is reported under "hbm_equivalent", never as frac.
  * two-pass kernels move the algorithmic bytes: bound = "hbm".
  "traffic" = HBM bytes per launch from separate FETCH_SIZE / WRITE_SIZE passes (counter
  unit KiB, FETCH doubled for gfx950 as the guide prescribes), taken by child processes
  of this run at N=1 before the parent touches the GPU ("counters": "live"); when
  rocprofv3 is unavailable the committed pass under profiles/ is used and named.
"cpu_baseline" times the CPU oracle (a port of the reference's dec(),
oracle/ldpc_oracle.c) on a bounded sample of the same workload: one process per core
(cores = P) and the one-core figure.
"""
import argparse
import glob
import json
import math
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CODE = os.path.join(ROOT, "tests", "golden", "codes", "peg_16000_3_0.9.txt")
HBM_PEAK_GBS = 8000.0                     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_GINST = 1024 * 2.4 / 2.0        # 256 CUs x 4 SIMD-32, wave64 instruction = 2 cycles, 2.4 GHz
SEED = 0x1DBC0001
COMMITTED_COUNTERS = os.path.join(ROOT, "profiles", "r03_counters.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=65536, help="frames per step per GPU")
    ap.add_argument("--alg", default="q88", choices=["q88", "f32", "f64"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--p", type=float, default=0.02)
    ap.add_argument("--code", default=CODE)
    ap.add_argument("--early-exit", type=int, default=0)
    ap.add_argument("--cn-variant", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel hipEvents")
    ap.add_argument("--no-secondary", action="store_true", help="skip the floating-point leg reported under 'secondary'")
    ap.add_argument("--no-pmc", action="store_true", help="no rocprofv3 counter passes (use the committed ones)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--dry", action="store_true",
                    help="plumbing check without a GPU: gloo ranks, no decoding (tests/test_bench_launcher.py)")
    ap.add_argument("--save-counters", default=None,
                    help="write the live counter passes to this JSON (the committed fallback, profiles/r03_counters.json)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle spot-check of the last timed step")
    ap.add_argument("--verify-frames", type=int, default=32)
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the short legs on BASELINE.json's other configs (reported under 'configs')")
    ap.add_argument("--config-child", type=int, default=-1, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ------------------------------------------------------------------ launcher

def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """N fresh rank processes of this script under torch.distributed.run (nothing in this
    process has touched the GPU); rank 0's JSON line is passed through."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if r.returncode != 0 or line is None:
        print(f"bench.py: the {args.gpus}-rank launch failed (exit {r.returncode})", file=sys.stderr)
        raise SystemExit(r.returncode or 1)
    print(line, flush=True)


# ------------------------------------------------------- counter passes (N=1)

def pmc_sum(csv_path, kernel_substr):
    """Per-counter sums and dispatch count over the dispatches of the kernels whose name holds
    kernel_substr (rocprofv3 --pmc ... --output-format csv)."""
    import csv
    acc, disp = {}, set()
    with open(csv_path) as f:
        for r in csv.DictReader(f):
            if kernel_substr not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    return acc, len(disp)


PMC_PASSES = (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
              ("sq", ["SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                      "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES"]))


def live_counters(args, kernel_substr, child=None, passes=None):
    """One rocprofv3 --pmc pass per counter group over `bench.py --pmc-child` (the same
    workload, one launch), as children of this process before it touches the GPU.
    -> {counter: value per launch} or None."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    out = {}
    base = tempfile.mkdtemp(prefix="ldpc_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    if child is None:
        child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--gpus", "1", "--steps", "1", "--warmup", "0",
                 "--frames", str(args.frames), "--alg", args.alg, "--iters", str(args.iters), "--p", str(args.p),
                 "--code", args.code, "--early-exit", str(args.early_exit), "--cn-variant", str(args.cn_variant)]
    try:
        for tag, counters in (passes or PMC_PASSES):
            d = os.path.join(base, tag)
            cmd = [exe, "--pmc"] + counters + ["-d", d, "-o", "p", "--output-format", "csv", "--"] + child
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                   text=True, timeout=150)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 pass '{tag}' timed out"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 pass '{tag}' failed (exit {r.returncode})"
            acc, nd = pmc_sum(files[0], kernel_substr)
            if nd == 0:
                return None, f"no dispatch of a '{kernel_substr}' kernel in pass '{tag}'"
            for k, v in acc.items():
                out[k] = v / nd
    finally:
        shutil.rmtree(base, ignore_errors=True)
    return out, "live"


def save_counters(args, counters):
    """Merges this run's per-launch counters into the committed fallback file."""
    doc = {}
    if os.path.exists(args.save_counters):
        with open(args.save_counters) as f:
            doc = json.load(f)
    doc["_comment"] = ("per-launch sums of rocprofv3 --pmc passes over `bench.py --pmc-child` (one launch of the bench "
                       "workload): FETCH_SIZE / WRITE_SIZE in KiB (FETCH to be doubled on gfx950), SQ_* as counted; "
                       "written by `bench.py --save-counters`; bench.py reads it only when a live pass is impossible")
    doc[args.alg] = {"workload": f"peg_16000_3_0.9 {args.alg} {args.frames} frames/launch {args.iters} iterations",
                     "frames_per_launch": args.frames, "per_launch": {k: float(v) for k, v in sorted(counters.items())}}
    with open(args.save_counters, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")


def committed_counters(args):
    if args.code != CODE or args.iters != 10 or args.cn_variant != 0 or not os.path.exists(COMMITTED_COUNTERS):
        return None
    with open(COMMITTED_COUNTERS) as f:
        t = json.load(f).get(args.alg)
    if not t:
        return None
    scale = args.frames / t["frames_per_launch"]
    return {k: v * scale for k, v in t["per_launch"].items()}


# ------------------------------------------------------------- cpu baseline

def _cpu_worker(job):
    """Frames of the bench workload through the CPU oracle for ~seconds; -> (frames, seconds)."""
    code_path, alg, iters, p, seconds, seed = job
    from oracle.bindings import Oracle
    ora = Oracle()
    oc = ora.load(code_path)
    n, m = oc.n, oc.m
    rng = np.random.default_rng(seed)
    L = math.log((1 - p) / p)
    Lq = round(L * 256) / 256.0
    s = np.zeros(m, np.int32)
    fn = (lambda q: ora.dec_minsum_fixed(oc, q, s, iters, None, m + 1)) if alg == "q88" else \
         (lambda q: ora.dec_minsum(oc, q, s, iters))
    frames, spent = 0, 0.0
    while spent < seconds or frames < 4:
        q0 = np.where(rng.random(n) < p, -Lq, Lq).astype(np.float64)
        t = time.perf_counter()
        fn(q0)
        spent += time.perf_counter() - t
        frames += 1
    return frames, spent


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, K_info):
    """The CPU oracle (port of the reference dec(), O(dc^2) check node and all) on a bounded
    sample of the same workload: one core, then one process per core of this process's share
    of the host (the reference scales out the same way: one single-threaded worker per core,
    test_mex_B_float.m:24-30)."""
    import multiprocessing as mp
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    P = max(1, min(share, int(os.environ.get("OMP_NUM_THREADS", share)), 64))
    f1, t1 = _cpu_worker((args.code, args.alg, args.iters, args.p, args.cpu_seconds * 0.4, SEED))
    one = K_info * f1 / t1 / 1e9
    jobs = [(args.code, args.alg, args.iters, args.p, args.cpu_seconds * 0.6, SEED + 1 + i) for i in range(P)]
    w0 = time.perf_counter()
    with mp.get_context("spawn").Pool(P) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - w0
    # aggregate rate = sum of the workers' own rates (each times only its dec() calls, as
    # SURVEY 8(d) prescribes: not initdec, not the random numbers)
    multi = sum(K_info * f / t for f, t in res) / 1e9
    frames = sum(f for f, _ in res)
    return {"value": round(multi, 6), "unit": "Gbit/s", "cores": P, "kind": "port", "cpu_model": cpu_model(),
            "note": "the port keeps the reference's O(dc^2) check node but walks index tables where the reference chases "
                    "pointer tables; the reference's own dec() (a mex source, not buildable here without stand-in headers) "
                    "was probed by the survey at 0.41 Mbit/s per Xeon core on this workload (BASELINE.md): the port is "
                    "about 4x kinder to the CPU than the reference, so GPU/CPU ratios against it understate the reference's gap",
            "one_core": {"value": round(one, 6), "frames": f1, "seconds": round(t1, 2)},
            "sample": f"{frames} frames of the same workload through oracle/ldpc_oracle.c (restatement of the "
                      f"reference dec(), gcc -O2), {P} single-threaded processes side by side for "
                      f"{args.cpu_seconds * 0.6:.0f} s each ({wall:.1f} s wall incl. start-up); one_core: {f1} frames, "
                      f"one process alone"}


# ------------------------------------------------ oracle spot-check (checker)

def verify_frames(code_path, alg, llr_t, bits_t, iters_t, status_t, max_iter, nver):
    """The CPU oracle as the CHECKER of what was just timed (never as the thing measured): `nver`
    frames spread over the last step's batch -- the first, the last, evenly in between -- are
    copied to the host with the decoder's outputs and decoded again by oracle/ldpc_oracle.c
    (ora_dec_minsum_fixed / ora_dec_minsum_f32 / ora_dec_minsum); decoded bits, iteration count
    and status must be equal.  -> (frames checked, frames that differ)."""
    import torch
    from oracle.bindings import Oracle
    F = int(llr_t.shape[0])
    idx = sorted(set(np.linspace(0, F - 1, min(nver, F)).astype(np.int64).tolist()))
    sel = torch.tensor(idx, device=llr_t.device)
    llr = llr_t[sel].cpu().numpy()
    bits = bits_t[sel].cpu().numpy()
    iters = iters_t[sel].cpu().numpy()
    status = status_t[sel].cpu().numpy()
    ora = Oracle()
    oc = ora.load(code_path)
    s = np.zeros(oc.m, np.int32)
    bad = 0
    for k in range(len(idx)):
        if alg == "q88":
            r = ora.dec_minsum_fixed(oc, llr[k].astype(np.float64) / 256.0, s, max_iter, None, oc.m + 1)
        elif alg == "f32":
            r = ora.dec_minsum_f32(oc, llr[k], s, max_iter)
        else:
            r = ora.dec_minsum(oc, llr[k], s, max_iter)
        same = r["rc"] == status[k] and r["iters"] == iters[k] and np.array_equal(r["bits"], bits[k])
        bad += 0 if same else 1
    return len(idx), bad


# ------------------------------------------- the other configs of BASELINE.json

CODES_DIR = os.path.join(ROOT, "tests", "golden", "codes")
# Short legs, one per remaining config, run after the headline (N = 1 only): the device Monte-Carlo
# loop (source, enc(), channel, dec(), classification on the GPU) for the binary ones, resident
# random priors for GF(4).  "fused": the frame-resident kernels (vector-issue bound, instruction
# count from one live rocprofv3 pass over `--config-child i`); the others are two-pass (HBM bound,
# algorithmic bytes per launch over the launch time).
CONFIG_LEGS = [
    dict(id="configs[0]", code="204.33.484.txt", alg="MINSUM_F64", dt="F64", channel="AWGN", ebn0=2.0, iters=50,
         frames=1 << 20, batch=1 << 18, fused=True, b=8,
         what="MacKay (204,102) min-sum in the reference's binary64, BPSK/AWGN Eb/N0 = 2 dB, <= 50 iterations, early exit"),
    dict(id="configs[2]/SPA_F64", code="dvb_H_0.5.txt", alg="SPA_F64", dt="F64", channel="AWGN", ebn0=1.0, iters=50,
         frames=4096, batch=4096, fused=False, b=8, issue_kernel="cn_reg_kernel",
         issue_note="a binary64 add / multiply / fma occupies the vector unit for 4 cycles against the 2 of the nominal peak: "
                    "a stream of them tops out at 0.5 of it",
         what="DVB-S2 rate 1/2 (n = 64800) sum-product in the reference's binary64, Eb/N0 = 1.0 dB, <= 50 iterations, early exit"),
    dict(id="configs[2]/SPA_TABLE_Q14", code="dvb_H_0.5.txt", alg="SPA_TABLE_Q14", dt="I32", channel="AWGN", ebn0=1.0,
         iters=50, frames=4096, batch=4096, fused=False, b=4,
         what="DVB-S2 rate 1/2 table sum-product (test_code_B), Eb/N0 = 1.0 dB, <= 50 iterations, early exit"),
    dict(id="configs[3]", code="32000.2240.3.105.txt", alg="MINSUM_Q88", dt="I16", channel="BSC_EXACT", p=13.0 / 32000,
         iters=200, frames=1 << 16, batch=1 << 16, fused=True, b=2,
         what="32000.2240.3.105 Q8.8 min-sum, exactly 13 flips per frame (FER floor region), <= 200 iterations, syndrome early exit"),
    dict(id="configs[3]/MINSUM_F32", code="32000.2240.3.105.txt", alg="MINSUM_F32", dt="F32", channel="BSC_EXACT", p=13.0 / 32000,
         iters=200, frames=1 << 16, batch=1 << 16, fused=True, b=4,
         what="32000.2240.3.105 binary32 min-sum (wide-row frame-resident kernel), exactly 13 flips per frame, <= 200 iterations, "
              "syndrome early exit"),
    dict(id="configs[3]/MINSUM_F64", code="32000.2240.3.105.txt", alg="MINSUM_F64", dt="F64", channel="BSC_EXACT", p=13.0 / 32000,
         iters=200, frames=1 << 14, batch=1 << 14, fused=False, b=8,
         what="32000.2240.3.105 min-sum in the reference's binary64 (256 KB of sums per frame: two-pass engine, spans sized to the "
              "Infinity Cache), exactly 13 flips per frame, <= 200 iterations, syndrome early exit"),
    dict(id="configs[2]/MINSUM_F32", code="dvb_H_0.5.txt", alg="MINSUM_F32", dt="F32", channel="AWGN", ebn0=2.4, iters=50,
         frames=4096, batch=4096, fused=True, kernel="stair_f32", b=4,
         what="DVB-S2 rate 1/2 binary32 min-sum (staircase frame-resident kernel), Eb/N0 = 2.4 dB, <= 50 iterations, early exit",
         note="the check pass is bound by vector issue, the rebuild of the sums by its 127 barrier steps; its per-workgroup record rows move 14.3 MB per frame at 10 iterations ("
              "profiles/r03_stair_f32_counters.txt)"),
    dict(id="configs[4]", code="q4.sp.9000.6000.4500.1", alg="GF2P_WHT", dt="I32", channel=None, iters=10, frames=2048,
         batch=2048, fused=False, b=4, issue_kernel="gf2p_cn",
         what="GF(4) q4.sp.9000.6000.4500.1 Walsh-Hadamard check nodes, 10 iterations executed, random priors resident"),
]


def run_config_leg(i, local_rank, child=False):
    """One leg of CONFIG_LEGS on this rank's GPU -> its record (child: one untimed run for the profiler)."""
    import torch
    import ldpc_amd
    from ldpc_amd import ALG, CH, DT
    from ldpc_amd.sweep import ebn0_to_sigma
    leg = CONFIG_LEGS[i]
    code = ldpc_amd.Code.from_alist(os.path.join(CODES_DIR, leg["code"]))
    n, m, E, q = code.n, code.m, code.E, code.q
    frames = leg["frames"] if not child else min(leg["frames"], leg["batch"])
    dec = ldpc_amd.Decoder(code, getattr(ALG, leg["alg"]), device=local_rank, max_frames=leg["batch"],
                           early_exit=leg["channel"] is not None)
    dt = getattr(DT, leg["dt"])
    rec = {"config": leg["id"], "workload": leg["what"], "alg": leg["alg"], "frames": frames, "iterations": leg["iters"]}
    if "note" in leg:
        rec["note"] = leg["note"]
    if leg["channel"] is None:                              # GF(4): resident random priors, all iterations executed
        dev = torch.device("cuda", local_rank)
        gen = torch.Generator(device=dev)
        gen.manual_seed(3)
        u = torch.rand((frames, n, q), device=dev, generator=gen)
        llr = (torch.log2(u / u.sum(dim=2, keepdim=True)) * 16384).to(torch.int32).contiguous()
        bits = torch.empty((frames, n), dtype=torch.uint8, device=dev)
        status = torch.empty(frames, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def run():
            dec.decode_device(frames, llr.data_ptr(), dt, leg["iters"], None, bits.data_ptr(), None, None,
                              status.data_ptr(), None, st)
            torch.cuda.synchronize()
            return None
        info_bits = (n - m) * math.log2(q)
    else:
        mc = ldpc_amd.MonteCarlo(code, dec, device=local_rank)
        par = ebn0_to_sigma(leg["ebn0"], (n - m) / n) if leg["channel"] == "AWGN" else leg["p"]

        def run():
            return mc.run(frames, getattr(CH, leg["channel"]), par, dt, leg["iters"], seed=SEED)
        info_bits = n - m
    run()                                                   # warm-up (first launches, lazy allocations)
    if child:
        return None
    dec.profile(True)
    t0 = time.perf_counter()
    cnt = run()
    sec = time.perf_counter() - t0
    ms_k, cnt_k = dec.profile_read()
    fl_k = dec.profile_frames()
    dec.profile(False)
    rec.update({"value": round(info_bits * frames / sec / 1e9, 4), "unit": "Gbit/s", "seconds": round(sec, 4),
                "Msymbols_per_s": round(n * frames / sec / 1e6, 1)})
    if cnt is not None:
        rec.update({"fer": round(cnt["not_converged"] / max(1, cnt["frames"]), 6), "undetected": cnt["undetected"],
                    "avg_iterations": round(cnt["iterations"] / max(1, cnt["frames"]), 3),
                    "includes": "source, enc(), channel, dec() and classification, all on the device"})
    names = ["variable pass", "check pass", "frame-resident whole-decode kernel", "other"]
    rec["_check_pass_avg_ms"] = float(ms_k[1]) / int(cnt_k[1]) if cnt_k[1] else None
    if cnt_k.sum() > 0:
        k = int(np.argmax(ms_k))
        avg_ms = float(ms_k[k]) / max(1, int(cnt_k[k]))
        b = leg["b"] * (q if q > 2 else 1)
        rec["kernel"] = names[k]
        rec["kernel_ms"] = {names[j]: round(float(ms_k[j]), 3) for j in range(4) if cnt_k[j]}
        if k in (0, 1):
            # bytes of every launch of the class: each launch priced by the frames it worked on
            # (ldpc_decoder_profile_frames: the second stage of a two-stage early exit holds fewer)
            byt = ((2 * E * b + n * b) if k == 0 else 2 * E * b) * int(fl_k[k])
            ach = byt / (float(ms_k[k]) * 1e-3) / 1e9
            rec["roofline"] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg_ms, 4),
                               "launches": int(cnt_k[k]), "algorithmic_bytes_per_launch": byt // max(1, int(cnt_k[k])),
                               "traffic": None,
                               "figure": "SURVEY 8(d) bytes of that pass x the frames each launch worked on, summed over "
                                         "the launches / their summed time"}
        else:
            rec["roofline"] = {"bound": "valu", "achieved": None, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
                               "frac": None, "avg_launch_ms": round(avg_ms, 4), "launches": int(cnt_k[k]), "traffic": None}
    return rec


CONFIG_PMC_PASS = (("sq", ["SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]),)


def config_counters(args):
    """Live vector-instruction counts of the legs whose dominant kernel is frame-resident:
    one rocprofv3 pass each over `bench.py --config-child i` (before the parent touches the GPU)."""
    out = {}
    for i, leg in enumerate(CONFIG_LEGS):
        if not leg["fused"] and "issue_kernel" not in leg:
            continue
        child = [sys.executable, os.path.abspath(__file__), "--config-child", str(i)]
        # (two-pass legs whose check pass is bound by vector issue, not by its bytes: that kernel's own count)
        c, src = live_counters(args, leg.get("kernel", "fused_") if leg["fused"] else leg["issue_kernel"], child=child,
                               passes=CONFIG_PMC_PASS)
        out[i] = (c, src)
    return out


# --------------------------------------------------------------------- main

def dry_run(args, rank, world):
    """The rank plumbing without a GPU or the library: gloo group, barrier, the two
    all-reduces, one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rs = world
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([rank + 1, 1], dtype=torch.int64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        assert int(c[0]) == world * (world + 1) // 2
        rs = int(c[1])
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "dry run (no decoding)", "value": 0.0, "unit": "Gbit/s", "n_gpus": world,
                          "ranks_in_reduce": rs, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 3), "dry": True}), flush=True)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        launch_ranks(args)
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.dry:
        dry_run(args, rank, world)
        return
    if args.config_child >= 0:
        run_config_leg(args.config_child, local_rank, child=True)
        return

    b_msg = {"q88": 2, "f32": 4, "f64": 8}[args.alg]
    fused_possible = args.cn_variant == 0 and os.environ.get("LDPC_HIP_FUSED", "1") != "0"

    # Legs that start child processes come first: nothing here has touched the GPU yet.
    counters, counters_src, cpu, cfg_counters = None, None, None, {}
    if rank == 0 and world == 1 and not args.pmc_child:
        if not args.no_pmc:
            counters, counters_src = live_counters(args, "fused_" if fused_possible else "cn_")
        if counters is not None and args.save_counters:
            save_counters(args, counters)
        if counters is None:
            why = counters_src
            counters = committed_counters(args)
            counters_src = None if counters is None else \
                f"committed pass {os.path.relpath(COMMITTED_COUNTERS, ROOT)}" + (f" ({why})" if why else "")
        if not args.no_cpu:
            code_n_m = np.loadtxt(args.code, max_rows=1, dtype=int)
            cpu = cpu_baseline(args, int(code_n_m[0] - code_n_m[1]))
        if not args.no_configs and not args.no_pmc:
            cfg_counters = config_counters(args)
    elif rank == 0 and not args.pmc_child:
        # N > 1: no child processes beside the ranks; the instruction count per launch is the
        # committed one-GPU pass of the same workload (every rank runs that same launch)
        counters = committed_counters(args)
        counters_src = None if counters is None else \
            f"committed pass {os.path.relpath(COMMITTED_COUNTERS, ROOT)} (N > 1: no live pass)"

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libldpc_hip has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LDPC_BENCH_FORCE_DIST=1: take the RCCL path with one rank too (rehearsal on a one-GPU box under
    # `python -m torch.distributed.run --nproc-per-node 1 bench.py`)
    use_dist = world > 1 or os.environ.get("LDPC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == args.gpus

    import ldpc_amd
    from ldpc_amd import ALG, DT

    code = ldpc_amd.Code.from_alist(args.code)
    n, m, E = code.n, code.m, code.E
    K_info = n - m
    F = args.frames
    alg = {"q88": ALG.MINSUM_Q88, "f32": ALG.MINSUM_F32, "f64": ALG.MINSUM_F64}[args.alg]
    dec = ldpc_amd.Decoder(code, alg, device=local_rank, max_frames=F, early_exit=bool(args.early_exit),
                           cn_variant=args.cn_variant)

    # synthetic frames, generated on the device (counter-based Philox stream per rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED + rank)
    L = math.log((1 - args.p) / args.p)
    Lraw = int(round(L * 256))
    flips = torch.rand((F, n), device=dev, generator=gen) < args.p
    if args.alg == "q88":
        llr = torch.where(flips, -Lraw, Lraw).to(torch.int16).contiguous()
        dt = DT.I16
    elif args.alg == "f32":
        llr = torch.where(flips, -L, L).to(torch.float32).contiguous()
        dt = DT.F32
    else:
        llr = torch.where(flips, -L, L).to(torch.float64).contiguous()
        dt = DT.F64
    del flips
    bits = torch.empty((F, n), dtype=torch.uint8, device=dev)
    iters = torch.empty(F, dtype=torch.int32, device=dev)
    status = torch.empty(F, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        dec.decode_device(F, llr.data_ptr(), dt, args.iters, None, bits.data_ptr(), iters.data_ptr(), None,
                          status.data_ptr(), None, stream)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_profile:
        dec.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ms_k, cnt_k, fl_k = (np.zeros(4), np.zeros(4, np.int32), np.zeros(4, np.int64))
    if not args.no_profile:
        ms_k, cnt_k = dec.profile_read()
        fl_k = dec.profile_frames()
        dec.profile(False)
    if args.pmc_child:
        return

    # the oracle as the checker of the last timed step (rank 0): a number whose frames differ from
    # the reference's results is no number
    verified = None
    if rank == 0 and not args.no_verify:
        nv, bad = verify_frames(args.code, args.alg, llr, bits, iters, status, args.iters, args.verify_frames)
        if bad:
            raise SystemExit(f"bench.py: {bad} of {nv} frames of the last timed step differ from the CPU oracle")
        verified = nv

    # the path's only collective: error counters (and the max of the elapsed time)
    fails = int((status != 0).sum().item())
    undet = int(((status == 0) & (bits.any(dim=1))).sum().item())
    avg_it = float(iters.float().mean().item())
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([fails, undet, 1], dtype=torch.int64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        fails, undet = int(c[0].item()), int(c[1].item())
        assert int(c[2].item()) == world, "a rank is missing from the counter reduce"

    # configs[1] names both number formats: the float min-sum in the reference's own
    # precision (binary64, test_code_min_sum_B.c) is timed the same way on the same frames,
    # right after, and reported under "secondary"
    secondary = None
    if args.alg == "q88" and not args.no_secondary:
        del dec
        sF = min(F, 16384)
        dec2 = ldpc_amd.Decoder(code, ALG.MINSUM_F64, device=local_rank, max_frames=sF, early_exit=bool(args.early_exit))
        llr2 = (llr[:sF].to(torch.float64) * (L / Lraw)).contiguous()
        s_steps = max(1, args.steps // 4)

        def step2():
            dec2.decode_device(sF, llr2.data_ptr(), DT.F64, args.iters, None, bits.data_ptr(), iters.data_ptr(), None,
                               status.data_ptr(), None, stream)
        step2()
        fence()
        t0 = time.perf_counter()
        for _ in range(s_steps):
            step2()
        fence()
        el2 = time.perf_counter() - t0
        v2 = None
        if rank == 0 and not args.no_verify:
            nv, bad = verify_frames(args.code, "f64", llr2, bits[:sF], iters[:sF], status[:sF], args.iters,
                                    max(4, args.verify_frames // 2))
            if bad:
                raise SystemExit(f"bench.py: {bad} of {nv} frames of the MINSUM_F64 leg differ from the CPU oracle")
            v2 = nv
        if use_dist:
            t = torch.tensor([el2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        secondary = {"alg": "MINSUM_F64 (the reference's double min-sum, test_code_min_sum_B.c:137-208, bit-exact)",
                     "value": round(K_info * sF * s_steps * world / el2 / 1e9, 3), "unit": "Gbit/s",
                     "frames_per_step_per_gpu": sF, "steps": s_steps, "ms_per_step": round(el2 / s_steps * 1e3, 3),
                     "dtype": "f64", "avg_iterations": round(float(iters[:sF].float().mean().item()), 3),
                     "verified_frames": v2}
        del dec2, llr2

    # BASELINE.json's other configs, a short leg each (one GPU; the scaling runs time the headline only)
    configs = None
    if rank == 0 and world == 1 and not args.no_configs:
        if "dec" in dir():
            del dec
        del llr, bits
        torch.cuda.empty_cache()
        configs = []
        for i in range(len(CONFIG_LEGS)):
            try:
                rec = run_config_leg(i, local_rank)
            except Exception as e:                         # a leg that fails must not take the headline with it
                rec = {"config": CONFIG_LEGS[i]["id"], "error": f"{type(e).__name__}: {e}"}
            c, src = cfg_counters.get(i, (None, None))
            rl = rec.get("roofline")
            cn_ms = rec.pop("_check_pass_avg_ms", None)
            if rl and rl["bound"] == "hbm" and "issue_kernel" in CONFIG_LEGS[i]:
                # the check pass of this leg against the vector-issue ceiling, beside the byte figure above
                iss = {"kernel": CONFIG_LEGS[i]["issue_kernel"], "counters": src, "peak": VALU_PEAK_GINST,
                       "unit": "G wave-instructions/s", "avg_launch_ms": None if cn_ms is None else round(cn_ms, 4)}
                if c and c.get("SQ_INSTS_VALU") and cn_ms:
                    ach = c["SQ_INSTS_VALU"] / (cn_ms * 1e-3) / 1e9
                    iss.update({"valu_wave_instructions_per_launch": int(c["SQ_INSTS_VALU"]), "achieved": round(ach, 1),
                                "frac": round(ach / VALU_PEAK_GINST, 4)})
                    if c.get("SQ_WAVE_CYCLES"):
                        iss["wait_any_share"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
                if "issue_note" in CONFIG_LEGS[i]:
                    iss["note"] = CONFIG_LEGS[i]["issue_note"]
                rl["check_pass_vector_issue"] = iss
            if rl and rl["bound"] == "valu":
                rl["counters"] = src
                if c and c.get("SQ_INSTS_VALU"):
                    ach = c["SQ_INSTS_VALU"] / (rl["avg_launch_ms"] * 1e-3) / 1e9
                    rl.update({"achieved": round(ach, 1), "frac": round(ach / VALU_PEAK_GINST, 4),
                               "valu_wave_instructions_per_launch": int(c["SQ_INSTS_VALU"])})
                    if c.get("SQ_WAVE_CYCLES"):
                        rl["wait_any_share"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
            configs.append(rec)

    if rank == 0:
        total_frames = F * args.steps * world
        value = K_info * total_frames / elapsed / 1e9
        names = ["vn_kernel", "cn_kernel", "fused_decode_kernel", "other"]
        # frames per launch of each kernel class: F for the frame-resident kernels; the two-pass engine cuts a large
        # call into spans of frames (BinaryEngine::frames_per_span), so its launches are priced by the frames each
        # worked on (ldpc_decoder_profile_frames: summed over the launches)
        fpl = [float(fl_k[i]) / max(1, int(cnt_k[i])) if fl_k[i] else float(F) for i in range(4)]
        per_launch_bytes = [int((2 * E * b_msg + n * b_msg) * fpl[0]), int(2 * E * b_msg * fpl[1]),
                            int(args.iters * (4 * E * b_msg + n * b_msg) * fpl[2]), 0]
        roofline = None
        if cnt_k.sum() > 0:
            k = int(np.argmax(ms_k))
            avg_ms = ms_k[k] / max(1, cnt_k[k])
            sec = avg_ms * 1e-3
            ach_hbm = per_launch_bytes[k] / sec / 1e9
            traffic = None
            if counters and "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
                # KiB units; FETCH_SIZE tallies 128-B requests at 64 B on gfx950: doubled (MI355X_MICROARCH.md)
                traffic = int(round((2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024))
            common = {"kernel": names[k], "traffic": traffic, "avg_launch_ms": round(avg_ms, 4),
                      "launches": int(cnt_k[k]), "counters": counters_src,
                      "actual_hbm_GBps": None if traffic is None else round(traffic / sec / 1e9, 1),
                      "all_kernels_ms": {names[i]: round(float(ms_k[i]), 3) for i in range(4) if cnt_k[i]}}
            if names[k] == "fused_decode_kernel":
                insts = counters.get("SQ_INSTS_VALU") if counters else None
                ach = None if insts is None else insts / sec / 1e9
                roofline = {"bound": "valu", "achieved": None if ach is None else round(ach, 1),
                            "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
                            "frac": None if ach is None else round(ach / VALU_PEAK_GINST, 4),
                            "valu_wave_instructions_per_launch": None if insts is None else int(insts),
                            "figure": "vector ALU issue: SQ_INSTS_VALU per launch / launch time against 1024 SIMDs x "
                                      "2.4 GHz / 2 cycles per wave64 instruction; messages never leave LDS / VGPRs, "
                                      "so the HBM roofline does not bound this kernel",
                            "hbm_equivalent": {"algorithmic_bytes_per_launch": per_launch_bytes[k],
                                               "GBps": round(ach_hbm, 1), "over_hbm_peak": round(ach_hbm / HBM_PEAK_GBS, 4),
                                               "note": "SURVEY 8(d) two-pass algorithmic bytes / launch time: an "
                                                       "equivalence figure, NOT bandwidth moved and not a roofline "
                                                       "fraction"}}
                if counters and counters.get("SQ_WAVE_CYCLES"):
                    wc = counters["SQ_WAVE_CYCLES"]
                    roofline["sq"] = {c: round(counters[c] / wc, 4) for c in
                                      ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if c in counters}
                    roofline["sq"]["note"] = "shares of SQ_WAVE_CYCLES, from the profiled (lower-clocked) pass"
                    for c in ("SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
                        if c in counters:
                            roofline["sq"][c + "_per_launch"] = int(counters[c])
            else:
                roofline = {"bound": "hbm", "achieved": round(ach_hbm, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach_hbm / HBM_PEAK_GBS, 4),
                            "algorithmic_bytes_per_launch": per_launch_bytes[k],
                            "frames_per_launch": round(fpl[k], 1),
                            "figure": "algorithmic bytes per launch (SURVEY 8d: that pass's bytes per frame x the mean "
                                      "frames per launch) / mean launch time"}
            roofline.update(common)
        whole = args.iters * (4 * E * b_msg + n * b_msg) * total_frames / elapsed / 1e9 / world
        fmt = {"q88": "Q8.8 fixed-point", "f32": "binary32", "f64": "binary64"}[args.alg]
        out = {
            "metric": "decoded information throughput, N=16000 R=0.9 min-sum, 10 iterations",
            "value": round(value, 3), "unit": "Gbit/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"q88": "i16", "f32": "f32", "f64": "f64"}[args.alg], "data": "synthetic",
            "config": {"workload": f"peg_16000_3_0.9 {fmt} min-sum, {args.iters} iterations executed per frame, "
                                   f"BSC p={args.p}, all-zero codeword", "frames_per_step_per_gpu": F,
                       "total_frames": total_frames, "n": n, "m": m, "edges": E,
                       "early_exit": args.early_exit, "parallelism": f"frames sharded over {world} GPU(s)",
                       "ranks": world},
            "avg_iterations": round(avg_it, 3), "frame_failures": fails, "undetected": undet,
            "hbm_equivalent_GBps_per_gpu": round(whole, 1),
            "verified_frames": verified,
            "roofline": roofline, "cpu_baseline": cpu,
            "secondary": secondary, "configs": configs,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
