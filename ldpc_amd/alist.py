"""alist reading / writing with the call shapes of the reference's
matrix_alist_writer.py (matrix_to_alist :11-49, alist_to_mat :51-82).

The reader goes through libldpc_hip's loader (so it accepts every dialect the
C side accepts: padded or not, tabs, NB pairs) instead of the reference's
split(' ') parser, which needs the trailing-space style its own writer emits.
"""
import numpy as np

from .lib import Code


def matrix_to_alist(H, file_name):
    """Writes dense binary H[m][n] as a padded MacKay alist (1-based indices,
    rows/columns padded with 0 to the maximum weight, trailing space per line:
    byte-compatible with matrix_alist_writer.py:11-49)."""
    H = np.asarray(H)
    r, c = int(H.shape[0]), int(H.shape[1])
    colw = np.count_nonzero(H, axis=0)
    roww = np.count_nonzero(H, axis=1)
    cmax, rmax = int(colw.max()), int(roww.max())
    with open(file_name, "w") as f:
        f.write(f"{c} {r}\n{cmax} {rmax}\n")
        f.write("".join(f"{int(w)} " for w in colw) + "\n")
        f.write("".join(f"{int(w)} " for w in roww) + "\n")
        for i in range(c):
            idx = np.where(H[:, i] == 1)[0]
            f.write("".join(f"{j + 1} " for j in idx) + "0 " * (cmax - len(idx)) + "\n")
        for i in range(r):
            idx = np.where(H[i, :] == 1)[0]
            f.write("".join(f"{j + 1} " for j in idx) + "0 " * (rmax - len(idx)) + "\n")


def alist_to_mat(filename):
    """Dense H[m][n] (int) of an alist file; GF coefficients for NB files."""
    code = Code.from_alist(filename)
    rp, ci, hv = code.csr()
    H = np.zeros((code.m, code.n), dtype=int)
    for j in range(code.m):
        H[j, ci[rp[j]:rp[j + 1]]] = hv[rp[j]:rp[j + 1]]
    return H
