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


def _ints(path):
    with open(path) as f:
        return [int(t) for t in f.read().split()]


def _line(vals):
    # MATLAB's fprintf('%d ', v): every value followed by one blank
    return "".join(f"{int(v)} " for v in vals)


def sort_alist(file_name_r, file_name_w):
    """Matlab_LDPC_decoder/sort_alist.m: reads an UNPADDED binary alist, sorts the indices
    of every column and row list and writes them padded with zeros to the maximum weight.
    Byte-compatible with the .m file: every number is followed by a blank, and a list that
    needs no padding still gets the single blank MATLAB prints for fprintf('%d ', [])."""
    t = _ints(file_name_r)
    n, m, cmax, rmax = t[0], t[1], t[2], t[3]
    colw, roww = t[4:4 + n], t[4 + n:4 + n + m]
    at = 4 + n + m
    out = [_line([n, m]), _line([cmax, rmax]), _line(colw), _line(roww)]
    for wmax, weights in ((cmax, colw), (rmax, roww)):
        for w in weights:
            idx = sorted(t[at:at + w])
            at += w
            out.append(_line(idx) + (_line([0] * (wmax - w)) if wmax > w else " "))
    with open(file_name_w, "w") as f:
        f.write("\n".join(out) + "\n")


def matrix_to_alist_NB(H, file_name, q):
    """matrix_to_alist_NB.m: dense H[m][n] over GF(q) (0 = no edge) as a non-binary alist:
    header 'n m q', then per column / row the (index, coefficient) pairs in ascending index
    order, padded with '0 0 ' pairs to the maximum weight."""
    H = np.asarray(H)
    r, c = int(H.shape[0]), int(H.shape[1])
    colw = np.count_nonzero(H, axis=0)
    roww = np.count_nonzero(H, axis=1)
    cmax, rmax = int(colw.max()), int(roww.max())
    with open(file_name, "w") as f:
        f.write(f"{c} {r} {int(q)}\n{cmax} {rmax}\n")
        f.write(_line(colw) + "\n" + _line(roww) + "\n")
        for i in range(c):
            idx = np.nonzero(H[:, i])[0]
            f.write("".join(f"{j + 1} {int(H[j, i])} " for j in idx) + "0 0 " * (cmax - len(idx)) + "\n")
        for i in range(r):
            idx = np.nonzero(H[i, :])[0]
            f.write("".join(f"{j + 1} {int(H[i, j])} " for j in idx) + "0 0 " * (rmax - len(idx)) + "\n")


def alist_to_mat_NB(filename):
    """alist_to_mat_NB.m: dense H[m][n] of GF coefficients, built from the column lists
    alone (the .m file never reads the row lists)."""
    t = _ints(filename)
    n, m = t[0], t[1]
    cmax = t[3]
    colw = t[5:5 + n]
    at = 5 + n + m
    H = np.zeros((m, n), dtype=int)
    for i in range(n):
        for k in range(colw[i]):
            H[t[at + 2 * k] - 1, i] = t[at + 2 * k + 1]
        at += 2 * cmax
    return H


def normalise_alist(file_name_r, file_name_w, padded=True):
    """Rewrites a binary alist in one dialect: padded (what the reference's C loaders need,
    e.g. for the unpadded DVB files its fscanf loop misparses) or unpadded (what
    sort_alist.m reads).  Any dialect the library's loader accepts can be the input; index
    lists come out ascending."""
    code = Code.from_alist(file_name_r)
    if code.q != 2:
        raise ValueError("normalise_alist is for binary codes; use alist_to_mat_NB / matrix_to_alist_NB")
    rp, ci, _ = code.csr()
    cp, eo = code.csc()
    row_of_edge = np.repeat(np.arange(code.m), np.diff(rp))
    colw, roww = np.diff(cp), np.diff(rp)
    cmax, rmax = int(colw.max()), int(roww.max())
    with open(file_name_w, "w") as f:
        f.write(f"{code.n} {code.m}\n{cmax} {rmax}\n")
        f.write(_line(colw) + "\n" + _line(roww) + "\n")
        for v in range(code.n):
            rows = sorted(int(row_of_edge[e]) + 1 for e in eo[cp[v]:cp[v + 1]])
            f.write(_line(rows) + ("0 " * (cmax - len(rows)) if padded else "") + "\n")
        for j in range(code.m):
            cols = sorted(int(x) + 1 for x in ci[rp[j]:rp[j + 1]])
            f.write(_line(cols) + ("0 " * (rmax - len(cols)) if padded else "") + "\n")
