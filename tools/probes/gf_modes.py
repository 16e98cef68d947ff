"""Which GF(2^p) kernel family disagrees with the oracle on a random code (debugging aid):
python tools/probes/gf_modes.py q m n rlo rhi colcap p"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("tg", os.path.join(ROOT, "tests", "test_gpu_gf.py"))
tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
import ldpc_amd as hip
from oracle.bindings import Oracle
q, m, n, rlo, rhi, cc = map(int, sys.argv[1:7]); p = float(sys.argv[7])
ora = Oracle()
rng = np.random.default_rng(q * 1000 + m)
H = tg._random_nb_code(rng, q, m, n, (rlo, rhi), cc)
path = os.path.join(tempfile.mkdtemp(), "nb.alist")
hip.alist.matrix_to_alist_NB(H, path, q)
oc = ora.load(path)
F, it = 37, 12
x, s, logfna, logp = tg._wht_frames(ora, oc, q, F, p, 5)
ref = [ora.dec_gf2p_wht(oc, int(np.log2(q)), logfna[f], s[f], it) for f in range(F)]
for mode in ("0", "2", "1"):
    os.environ["LDPC_HIP_GF_LDS_TABLES"] = mode
    code = hip.Code.from_alist(path)
    for its in (1, 2, it):
        r2 = [ora.dec_gf2p_wht(oc, int(np.log2(q)), logfna[f], s[f], its) for f in range(F)]
        dec = hip.Decoder(code, hip.ALG.GF2P_WHT, max_frames=64)
        out = dec.decode(logfna.astype(np.int32), s.astype(np.uint8), its)
        bad_bits = [f for f in range(F) if not np.array_equal(out["bits"][f], r2[f]["x"])]
        bad_st = [f for f in range(F) if (0 if r2[f]["rc"] == 0 else 1) != out["status"][f]]
        bad_it = [f for f in range(F) if r2[f]["iters"] != out["iters"][f]]
        print(f"mode {mode} max_iter {its}: frames with different symbols {bad_bits[:8]} status {bad_st[:8]} iters {bad_it[:8]}", flush=True)
