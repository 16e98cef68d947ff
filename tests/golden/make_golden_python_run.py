#!/usr/bin/env python3
"""Golden vectors for the Python caller's API shape: outputs of the reference's
own `parallel.run(H, LLR_init, max_iterations)` (parallel.py:291-314, the cleaned copy of
LDPC_decode.run, which itself is Python 2 and cannot be imported) on the (204,102) MacKay
code and on small_code.txt.  Run in the build container only (it imports /root/reference);
writes tests/golden/python_run.npz.  parallel.py does `from time import clock` (removed in
Python 3.8): time.clock is set to perf_counter before the import, as SURVEY 8c prescribes.

  python tests/golden/make_golden_python_run.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def main():
    import matplotlib
    matplotlib.use("Agg")
    time.clock = time.perf_counter
    sys.path.insert(0, REF)
    import parallel                                   # the reference module
    from ldpc_amd.alist import alist_to_mat
    out = {}
    rng = np.random.default_rng(20171)
    for tag, fname, ps, frames, maxit in (("c204", "204.33.484.txt", (0.03, 0.045, 0.06, 0.09), 4, 30),
                                          ("small", "small_code.txt", (0.05, 0.15), 4, 20)):
        H = np.asarray(alist_to_mat(os.path.join(HERE, "codes", fname)), dtype=int)
        m, n = H.shape
        llrs, cws, its = [], [], []
        for p in ps:
            for f in range(frames):
                flips = rng.random(n) < (0.0 if (f == 0 and p == ps[0]) else p)     # first frame: a codeword as is
                llr = (1 - 2 * flips.astype(int)) * np.log((1 - p) / p)
                # a 1-D array, as the reference's own callers pass (parallel.py:48,:68); an (n,1) array would be
                # accumulated into in place by `belief += ...` (:195-197, :211-213)
                cw, it, _secs = parallel.run(H, llr.copy(), maxit)
                llrs.append(llr)
                cws.append(np.asarray(cw).reshape(-1).astype(np.uint8))
                its.append(0 if it is True else (-1 if it is False else int(it)))
                print(tag, p, f, "iters", it, "weight", int(cws[-1].sum()), flush=True)
        out[tag + "_llr"] = np.stack(llrs)
        out[tag + "_cw"] = np.stack(cws)
        out[tag + "_iters"] = np.array(its, np.int32)       # 0: True sentinel, -1: False
        out[tag + "_maxit"] = np.int32(maxit)
    np.savez_compressed(os.path.join(HERE, "python_run.npz"), **out)
    print("written", os.path.join(HERE, "python_run.npz"))


if __name__ == "__main__":
    main()
