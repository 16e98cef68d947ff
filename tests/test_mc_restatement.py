"""CPU checks of oracle/mc_frontend.py (the numpy restatement the GPU Monte-Carlo tests
compare against): the generator is splitmix64, and the channel follows the reference's
bsc() / float_to_fix() / float2fix() (test_code_min_sum_B.c:41-67, _fixed.c:43-53,
test_code_B.c:36-43)."""
import numpy as np

from conftest import code_path
from oracle import mc_frontend as ref


def test_generator_is_splitmix64():
    # splitmix64 seeded with 0: published first outputs
    want = [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    got = [int(ref.draw(np.uint64(0), i)) for i in range(3)]
    assert got == want


def test_bsc_law_and_quantisers(oracle):
    oc = oracle.load(code_path("204.33.484.txt"))
    n = oc.n
    import ldpc_amd
    rp, ci, _ = ldpc_amd.Code.from_alist(code_path("204.33.484.txt")).csr()
    x, llr, s, d = ref.generate(n, rp, ci, 20, ref.BSC_EXACT, 0.031, ref.I16, 4)
    flips = int(n * 0.031 + 0.5)
    pp = flips / n
    assert (((d < 0) != (x == 1)).sum(axis=1) == flips).all()
    assert np.allclose(np.abs(d), np.log((1 - pp) / pp), rtol=0, atol=0)
    assert set(np.unique(np.abs(llr.astype(int)))) == {int(np.floor(np.log((1 - pp) / pp) * 256 + 0.5))}
    # enc(): the oracle's syndrome of the same words
    for f in range(20):
        assert np.array_equal(oracle.enc(oc, x[f].astype(np.int32)).astype(np.uint8), s[f])
    assert ref.float_to_fix_raw(np.array([0.5 / 256, -0.5 / 256, 200.0, -200.0, 1.49 / 256])).tolist() == [1, -1, 32512, -32768, 1]
    assert ref.float2fix14(np.array([0.5 / 16384, -0.5 / 16384, 1.0])).tolist() == [1, -1, 16384]
