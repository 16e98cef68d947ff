"""Where does SPA_F32 first produce a NaN on a given frame?  (diagnostic for tests/test_gpu_binary.py)"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import code_path, make_frames
import ldpc_amd as hip
from oracle.bindings import Oracle
ora = Oracle(); oc = ora.load(code_path("1057.244.3.txt"))
x, s, llr = make_frames(ora, oc, 530, 0.012 * 1.6, 14, "bsc")
f = 480
l32 = llr[f:f + 1].astype(np.float32)
code = hip.Code.from_alist(code_path("1057.244.3.txt"))
dec = hip.Decoder(code, hip.ALG.SPA_F32, max_frames=64, want_llr=True, early_exit=False)
for it in range(1, 12):
    o = dec.decode(l32, s[f:f + 1].astype(np.uint8), it, want=("bits", "iters", "status", "llr"))
    r = ora.dec_spa_f32(oc, l32[0], s[f], it)
    g, c = o["llr"][0].astype(np.float64), r["post"].astype(np.float64)
    fin = np.isfinite(g) & np.isfinite(c)
    rel = np.abs(g[fin] - c[fin]) / np.maximum(np.abs(c[fin]), 1e-3)
    print(f"it {it}: gpu status {o['status'][0]} iters {o['iters'][0]} nan {np.isnan(g).sum()} inf {np.isinf(g).sum()} zeros {(g == 0).sum()} min|post| {np.nanmin(np.abs(g)):.3g} max {np.nanmax(np.abs(g[np.isfinite(g)])):.4g}"
          f" | cpu rc {r['rc']} iters {r['iters']} nan {np.isnan(c).sum()} inf {np.isinf(c).sum()} zeros {(c == 0).sum()} min {np.nanmin(np.abs(c)):.3g} max {np.nanmax(np.abs(c[np.isfinite(c)])):.4g} | maxrel {rel.max():.3g} bitdiff {(o['bits'][0] != r['bits']).sum()}")
