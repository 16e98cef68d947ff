"""Alist tooling against the files the reference itself holds.

Fixtures (all under tests/golden/codes, copied data files of the reference):
  204.33.484.txt -> 204.33.484_sorted.txt   the input / output pair of sort_alist.m
  peg_4376_4_0.9_NB.txt                     written by matrix_to_alist_NB.m (GF(13))
  peg_16000_3_0.9.txt, small_code.txt       written in the matrix_to_alist.m / matrix_alist_writer.py style
"""
import filecmp

import numpy as np
import pytest

from conftest import code_path


def test_sort_alist_reproduces_the_reference_pair(hip, tmp_path):
    out = str(tmp_path / "sorted.txt")
    hip.alist.sort_alist(code_path("204.33.484.txt"), out)
    assert filecmp.cmp(out, code_path("204.33.484_sorted.txt"), shallow=False)


def test_nb_reader_writer_round_trip_is_byte_identical(hip, tmp_path):
    src = code_path("peg_4376_4_0.9_NB.txt")
    H = hip.alist.alist_to_mat_NB(src)
    assert H.shape == (282, 4376) and H.max() == 12 and np.count_nonzero(H) == 4 * 4376
    assert np.array_equal(H, hip.alist.alist_to_mat(src))            # the C loader reads the same matrix
    out = str(tmp_path / "nb.txt")
    hip.alist.matrix_to_alist_NB(H, out, 13)
    assert filecmp.cmp(out, src, shallow=False)


@pytest.mark.parametrize("name", ["peg_16000_3_0.9.txt", "small_code.txt"])
def test_binary_writer_round_trip_is_byte_identical(hip, tmp_path, name):
    H = hip.alist.alist_to_mat(code_path(name))
    out = str(tmp_path / "b.txt")
    hip.alist.matrix_to_alist(H, out)
    assert filecmp.cmp(out, code_path(name), shallow=False)


@pytest.mark.parametrize("name", ["dvb_H_0.5.txt", "1057.244.3.txt", "Frolov_1024_0.5.txt"])
def test_normaliser_keeps_the_matrix(hip, oracle, tmp_path, name):
    """padded <-> unpadded: same H either way; the padded form is what the reference's C
    loaders need (they misparse the shipped unpadded DVB file), so the CPU oracle's
    reference-dialect loader must read it too."""
    src = code_path(name)
    pad, unpad = str(tmp_path / "p.txt"), str(tmp_path / "u.txt")
    hip.alist.normalise_alist(src, pad, padded=True)
    hip.alist.normalise_alist(src, unpad, padded=False)
    a, b, c = (hip.Code.from_alist(p) for p in (src, pad, unpad))
    for x in (b, c):
        assert (x.n, x.m, x.E) == (a.n, a.m, a.E)
        ra, ca, _ = a.csr()
        rx, cx, _ = x.csr()
        assert np.array_equal(ra, rx)
        for j in range(a.m):
            assert sorted(ca[ra[j]:ra[j + 1]]) == sorted(cx[rx[j]:rx[j + 1]])
    oc = oracle.load(pad)
    assert (oc.n, oc.m) == (a.n, a.m)
    # unpadded -> sort_alist.m -> the padded, sorted form again
    again = str(tmp_path / "s.txt")
    hip.alist.sort_alist(unpad, again)
    d = hip.Code.from_alist(again)
    assert np.array_equal(d.csr()[1], b.csr()[1])
