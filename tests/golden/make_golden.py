#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ .  Runs only in the
build container (needs /root/reference and oracle/_ref built by oracle/Makefile);
nothing here travels to the GPU box except its outputs.

Fixtures (data only -- inputs and expected outputs):
  frolov_kat.npz            y_2048 / y_2048_success of Verilog_LDPC_decoder_tester.m:16-17
  verilog_success.txt.gz    the reference's stored trace verilog.txt (expected output
                            of test_code_min_sum_B_fixed on y_2048_success, LLR 11/256)
  debug_204_minsum.txt      the reference's stored one-iteration float min-sum dump
                            (root debug.txt, (204,102) code)
  ref_table_spa.npz         inputs + outputs of the compiled reference test_code_no_mex_B.c
                            dec() (oracle/_ref/libref_no_mex_B.so) on seeded frames
  ref_ntt_q4.npz, ref_ntt_q8.npz
                            inputs + outputs of the compiled reference GFq_LDPC_NTT.c
                            (main():681-695 experiments, seeds 1..3, deterministic LCG)
"""
import ctypes as C
import gzip
import hashlib
import os
import re
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
CODES = os.path.join(HERE, "codes")


def frolov():
    text = open(os.path.join(REF, "Verilog_LDPC_decoder_tester.m")).read()
    out = {}
    for name in ("y_2048", "y_2048_success"):
        mm = re.search(r"^%s\s*=\s*\[([^\]]*)\]" % name, text, re.M)
        out[name] = np.array([int(t) for t in mm.group(1).replace(",", " ").split()], np.uint8)
        assert out[name].size == 2048
    np.savez_compressed(os.path.join(HERE, "frolov_kat.npz"), llr_mag_raw=np.int32(11), max_iter=np.int32(30),
                        **out)
    raw = open(os.path.join(REF, "verilog.txt"), "rb").read()
    with gzip.GzipFile(os.path.join(HERE, "verilog_success.txt.gz"), "wb", mtime=0) as f:
        f.write(raw)
    open(os.path.join(HERE, "verilog_success.sha256"), "w").write(hashlib.sha256(raw).hexdigest() + "\n")
    shutil.copy(os.path.join(REF, "debug.txt"), os.path.join(HERE, "debug_204_minsum.txt"))
    os.chmod(os.path.join(HERE, "debug_204_minsum.txt"), 0o644)


def table_spa():
    from oracle.bindings import RefNoMexB
    ref = RefNoMexB()
    data = {}
    rng = np.random.default_rng(2024)
    for tag, fname, F, p, it in (("c204", "204.33.484.txt", 48, 0.06, 30), ("c4095", "4095.737.3.txt", 10, 0.012, 30),
                                 ("c1057", "1057.244.3.txt", 24, 0.02, 20)):
        # the reference's loader (test_code_no_mex_B.c:264-269) skips cmax tokens per column
        ref.load(os.path.join(CODES, fname))
        n, m = ref.n, ref.m
        x = rng.integers(0, 2, size=(F, n)).astype(np.int32)
        x[: F // 2] = 0
        q0 = np.zeros((F, n), np.int32)
        s = np.zeros((F, m), np.int32)
        bits = np.zeros((F, n), np.uint8)
        synd = np.zeros((F, m), np.uint8)
        rc = np.zeros(F, np.int32)
        iters = np.zeros(F, np.int32)
        L = np.log((1 - p) / p)
        for f in range(F):
            ref.lib.enc(x[f].ctypes.data_as(C.c_void_p), s[f].ctypes.data_as(C.c_void_p))
            y = x[f] ^ (rng.random(n) < p)
            q0[f] = [ref.float2fix((1 - 2 * int(b)) * L) for b in y]
            r = ref.dec(q0[f], s[f], it)
            bits[f], synd[f], rc[f], iters[f] = r["bits"], r["synd"], r["rc"], r["iters"]
        data.update({f"{tag}_q0": q0, f"{tag}_s": s.astype(np.uint8), f"{tag}_bits": bits, f"{tag}_synd": synd,
                     f"{tag}_rc": rc, f"{tag}_iters": iters, f"{tag}_maxit": np.int32(it),
                     f"{tag}_x": x.astype(np.uint8)})
        print(tag, "rc", rc.tolist(), "iters", iters.tolist())
    np.savez_compressed(os.path.join(HERE, "ref_table_spa.npz"), **data)


def ntt():
    from oracle.bindings import RefNTT
    for log2q, fname, channel, level, it in ((2, "q4.sp.9000.6000.4500.1", "bsc", 0.08, 50),
                                             (3, "q8.sp.6000.4000.3000.1", "lap", 0.5, 50)):
        ref = RefNTT(log2q)
        ref.load(os.path.join(CODES, fname))
        xs, ss, ls, xh, zh, rcs = [], [], [], [], [], []
        for seed in (1, 2, 3):
            x, s, y, logfna = ref.experiment(seed, channel, level)
            r = ref.dec(logfna, s, it)
            xs.append(x); ss.append(s); ls.append(logfna); xh.append(r["x"]); zh.append(r["z"]); rcs.append(r["rc"])
            print(fname, seed, "rc", r["rc"], "symbol errors left", int((r["x"] != x).sum()))
        # also a harder, non-convergent level so that the abort heuristic (:533-539) is pinned
        hard = {"bsc": 0.2, "lap": 1.2}[channel]
        x, s, y, logfna = ref.experiment(4, channel, hard)
        r = ref.dec(logfna, s, it)
        print(fname, "hard rc", r["rc"])
        xs.append(x); ss.append(s); ls.append(logfna); xh.append(r["x"]); zh.append(r["z"]); rcs.append(r["rc"])
        np.savez_compressed(os.path.join(HERE, f"ref_ntt_q{1 << log2q}.npz"), x=np.stack(xs).astype(np.uint8),
                            s=np.stack(ss).astype(np.uint8), logfna=np.stack(ls), xhat=np.stack(xh).astype(np.uint8),
                            zhat=np.stack(zh).astype(np.uint8), rc=np.array(rcs, np.int32), maxit=np.int32(it))


if __name__ == "__main__":
    frolov()
    table_spa()
    ntt()
    print("golden fixtures written to", HERE)
