#!/usr/bin/env python3
"""Test infrastructure (not collected by pytest): one-off stress comparison of the min-sum decoders (binary32, binary64, Q8.8; the default engine of every code) with the CPU checker on many
random frames (more than the test suite holds): mixed channels, exact +0.0 and -0.0 inputs, huge magnitudes,
non-zero syndromes.  python tests/stress_parity.py [frames-per-code]   (needs a GPU and the built checker)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import ldpc_amd as hip
    from oracle.bindings import Oracle
    from conftest import code_path, to_q88
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ora = Oracle()
    # code, AWGN sigma range (around the waterfall of min-sum), iterations
    cases = [("32000.2240.3.105.txt", 0.30, 0.44, 14), ("dvb_H_0.5.txt", 0.62, 0.84, 18), ("peg_16000_3_0.9.txt", 0.36, 0.48, 12),
             ("4095.737.3.txt", 0.42, 0.58, 14), ("1057.244.3.txt", 0.45, 0.62, 15), ("Frolov_1024_0.5.txt", 0.6, 0.9, 20),
             ("204.33.484.txt", 0.6, 0.9, 20)]
    algs = [("MINSUM_F32", np.float32), ("MINSUM_F64", np.float64), ("MINSUM_Q88", np.int16)]
    total_bad = 0
    for name, s_lo, s_hi, it in cases:
        path = code_path(name)
        if name.startswith("dvb"):
            path = "/tmp/stress_dvb_padded.txt"
            hip.alist.normalise_alist(code_path(name), path, padded=True)
        oc = ora.load(path)
        n, m = oc.n, oc.m
        rng = np.random.default_rng(2026)
        x = rng.integers(0, 2, size=(F, n)).astype(np.int32)
        x[: F // 3] = 0
        s = np.stack([ora.enc(oc, x[f]) for f in range(F)]).astype(np.uint8)
        sig = rng.uniform(s_lo, s_hi, size=(F, 1))
        llr = (2 * ((1.0 - 2.0 * x) + rng.normal(0, 1, size=(F, n)) * sig) / sig ** 2).astype(np.float32)
        for f in range(0, F, 7):
            idx = rng.choice(n, size=max(1, n // 50), replace=False)
            llr[f, idx] = np.float32(0.0) if f % 2 else np.float32(-0.0)
        for f in range(3, F, 11):
            llr[f] *= np.float32(rng.choice([25.0, 80.0, 1e4]))
        for f in range(5, F, 13):       # equal magnitudes: ties everywhere
            llr[f] = np.sign(llr[f]) * np.float32(2.5)
        code = hip.Code.from_alist(code_path(name))
        for alg, dt in algs:
            if dt == np.int16:
                grid, inp = to_q88(llr.astype(np.float64))
                fn = lambda f: ora.dec_minsum_fixed(oc, grid[f], s[f].astype(np.int32), it, None, m + 1)
            elif dt == np.float64:
                inp = llr.astype(np.float64)
                fn = lambda f: ora.dec_minsum(oc, inp[f], s[f].astype(np.int32), it)
            else:
                inp = llr
                fn = lambda f: ora.dec_minsum_f32(oc, inp[f], s[f].astype(np.int32), it)
            dec = hip.Decoder(code, getattr(hip.ALG, alg), max_frames=256, early_exit=True, want_llr=True)
            out = dec.decode(inp, s, it, want=("bits", "iters", "synd", "status", "llr"))
            bad = 0
            verd = [0, 0]
            for f in range(F):
                r = fn(f)
                verd[1 if r["rc"] else 0] += 1
                same = (r["rc"] == out["status"][f] and r["iters"] == out["iters"][f] and np.array_equal(r["bits"], out["bits"][f])
                        and np.array_equal(r["synd"], out["synd"][f]))
                if same and r["rc"] != 0:
                    same = np.array_equal(np.asarray(r["post"], np.float32), out["llr"][f])
                if not same:
                    bad += 1
                    print(f"  {name} {alg}: frame {f} differs (checker rc {r['rc']} it {r['iters']}, GPU rc {out['status'][f]} it {out['iters'][f]})")
            print(f"{name} {alg}: {F} frames, {verd[0]} converged / {verd[1]} not, {bad} differ from the checker", flush=True)
            total_bad += bad
            dec.close()
    if total_bad:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
