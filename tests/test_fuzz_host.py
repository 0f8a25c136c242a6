"""CPU: a bounded, seeded share of the randomized campaign of tests/host_emul/fuzz_cases.py — the entry points of the whole-library host build against the oracles on
shapes no hand-written case names: clouds of one point beside clouds of 2500, fewer supports than K, coincident points, lattices and planes (exactly tied
distances), shadow entries, hub targets, ignored labels, every contrast flavour.  Search results, sample sequences, sub-sampled points and transposed tables bit
for bit; the contrast loss and gradient within the kernels' 1e-4 contract."""
import pytest

from tests.host_emul import fuzz_cases as Z


@pytest.mark.parametrize("which,seed,cases", [("knn", 205, 4), ("radius", 102, 25), ("grid", 103, 100), ("subsample", 110, 200), ("pyramid", 111, 4), ("fps", 104, 3), ("transpose", 105, 20), ("cbl", 106, 80), ("gather", 107, 80), ("aggregation", 108, 40),
                                              ("attention", 109, 10)])
def test_random_cases_equal_the_oracles(which, seed, cases):
    assert Z.run(which, seed, cases) == 0
