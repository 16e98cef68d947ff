import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import code_path, make_frames
from test_gpu_binary import SPA_CASES, run_oracle
import ldpc_amd as hip
from oracle.bindings import Oracle
oracle = Oracle()
for (name, F, p, sigma, it) in SPA_CASES:
    for kind in ("bsc", "awgn"):
        for prec in ("f64", "f32"):
            MARGIN = {"f64": 1e-6, "f32": 1e-2}[prec]
            oc = oracle.load(code_path(name))
            x, s, llr = make_frames(oracle, oc, F, p * 1.6, 14, kind, sigma=sigma * 1.08)
            if prec == "f32":
                llr = llr.astype(np.float32); ref = run_oracle(oracle.dec_spa_f32, oc, llr, s, it); alg = hip.ALG.SPA_F32
            else:
                ref = run_oracle(oracle.dec_spa, oc, llr, s, it); alg = hip.ALG.SPA_F64
            code = hip.Code.from_alist(code_path(name))
            dec = hip.Decoder(code, alg, max_frames=256, want_llr=True)
            out = dec.decode(llr, s.astype(np.uint8), it, want=("bits", "iters", "synd", "status", "llr"))
            post = ref[4]
            nan = np.isnan(post).any(axis=1)
            minabs = np.abs(np.where(np.isnan(post), 1.0, post)).min(axis=1)
            safe = ~nan & (minabs > MARGIN)
            agree = (out["status"] == ref[3]) & (out["iters"] == ref[1]) & (out["bits"] == ref[0]).all(axis=1)
            print(f"{name} {kind} {prec}: F={F} nan={nan.sum()} safe={safe.sum()} agree_all={agree.sum()} agree_safe={agree[safe].sum()}/{safe.sum()} conv={int((ref[3]==0).sum())}")
            for f in np.where(~agree)[0]:
                print(f"   frame {f}: nan={nan[f]} minabs={minabs[f]:.3g} ref(rc,it)=({ref[3][f]},{ref[1][f]}) gpu=({out['status'][f]},{out['iters'][f]}) bitdiff={(out['bits'][f]!=ref[0][f]).sum()} gpu_nan={np.isnan(out['llr'][f]).any()}")
