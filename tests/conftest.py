import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CODES = os.path.join(ROOT, "tests", "golden", "codes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def code_path(name):
    return os.path.join(CODES, name)


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def hip():
    """The product library; GPU tests fail loudly when it is missing."""
    import ldpc_amd
    ldpc_amd.load_library()
    return ldpc_amd


def make_frames(ora, ocode, F, p, seed, kind="bsc", zero_frac=0.5, sigma=None):
    """Seeded synthetic frames: x (source), s = H x, LLRs.
    kind 'bsc': i.i.d. Bernoulli(p) flips, LLR = +-log((1-p)/p);
    kind 'awgn': BPSK + Gaussian noise, LLR = 2 y / sigma^2 (distinct magnitudes)."""
    rng = np.random.default_rng(seed)
    n, m = ocode.n, ocode.m
    x = rng.integers(0, 2, size=(F, n)).astype(np.int32)
    nz = int(F * zero_frac)
    x[:nz] = 0
    s = np.stack([ora.enc(ocode, x[f]) for f in range(F)]).astype(np.int32)
    if kind == "bsc":
        flips = (rng.random((F, n)) < p).astype(np.int32)
        y = x ^ flips
        L = np.log((1 - p) / p)
        llr = (1 - 2 * y) * L
    else:
        tx = 1.0 - 2.0 * x
        y = tx + rng.normal(0, sigma, size=(F, n))
        llr = 2 * y / sigma ** 2
    return x, s, llr.astype(np.float64)


def to_q88(llr):
    """float_to_fix() on an array -> values on the 1/256 grid (float64) and raw int16."""
    r = np.floor(np.abs(llr) * 256.0 + 0.5) * np.sign(llr)  # C round(): half away from zero (np.round goes to even)
    r = np.clip(r, -128 * 256, 127 * 256)
    return r / 256.0, r.astype(np.int16)
