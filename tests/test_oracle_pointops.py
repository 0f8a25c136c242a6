"""CPU: the oracle (oracle/pointops_oracle.c) against the golden vectors produced from the reference's
own kernel bodies (tests/golden/gen_pointops_goldens.py) and SURVEY.md §7's known answers."""
import math
import os

import numpy as np
import pytest

from tests import oracle_lib as O

G = os.path.join(os.path.dirname(__file__), "golden")


def cases(npz, prefix):
    names = sorted({k.split("/")[1] for k in npz.files if k.startswith(prefix + "/")})
    return names


KNN = np.load(os.path.join(G, "pointops_knn.npz"))
FPS = np.load(os.path.join(G, "pointops_fps.npz"))
K310 = np.load(os.path.join(G, "pointops_k3_k10.npz"))


@pytest.mark.parametrize("name", cases(KNN, "knn"))
def test_knn_matches_reference_bits(name):
    g = lambda f: KNN[f"knn/{name}/{f}"]
    idx, d2 = O.knnquery(int(g("k")), g("xyz"), g("new_xyz"), g("offset"), g("new_offset"))
    np.testing.assert_array_equal(idx, g("idx"))
    np.testing.assert_array_equal(d2.view(np.uint32), g("dist2").view(np.uint32))  # bit-exact floats


def test_knn_survey_known_answers():
    # SURVEY.md §7 hard part 1: query origin, supports with d2 = 5,5,1,1,5,5
    ka = np.float32([[1, 2, 0], [2, 1, 0], [1, 0, 0], [0, 1, 0], [2, -1, 0], [-1, 2, 0]])
    q = np.float32([[0, 0, 0]])
    expect = {2: [3, 2], 3: [3, 2, 1], 4: [2, 3, 1, 0], 5: [3, 2, 4, 1, 0]}
    for k, e in expect.items():
        idx, _ = O.knnquery(k, ka, q, [6], [1])
        assert idx[0].tolist() == e
    idx, d2 = O.knnquery(4, ka[:2], q, [2], [1])
    assert idx[0].tolist() == [1, 0, 0, 0]
    assert d2[0].tolist() == [5.0, 5.0, 1e10, 1e10]


@pytest.mark.parametrize("name", cases(FPS, "fps"))
def test_fps_matches_reference(name):
    g = lambda f: FPS[f"fps/{name}/{f}"]
    idx, tmp = O.furthestsampling(g("xyz"), g("offset"), g("new_offset"), int(g("n_max")))
    assert O.lib().oracle_ref_block_threads(int(g("n_max"))) == int(g("block"))
    np.testing.assert_array_equal(idx, g("idx"))
    np.testing.assert_array_equal(tmp.view(np.uint32), g("tmp_after").view(np.uint32))


def test_fps_block_size_rule():
    n = FPS["opt_n_threads/n"]
    t = FPS["opt_n_threads/threads"]
    ours = np.array([O.lib().oracle_ref_block_threads(int(v)) for v in n], np.int32)
    np.testing.assert_array_equal(ours, t)
    # cuda_utils.h:12 log(n)/log(2) truncation == floor(log2 n) on this libm for all n < 2^21
    bad = [v for v in range(1, 1 << 21) if int(math.log(float(v)) / math.log(2.0)) != v.bit_length() - 1]
    assert bad == []


def test_fps_tree_reduction_is_bitreversed_min():
    # the claim the oracle's FPS is built on: among threads tied at the max, the shared-memory tree
    # (sampling_cuda_kernel.cu:64-123) returns the one with the smallest bit-reversed thread id
    rng = np.random.default_rng(0)
    lib = O.lib()
    for B in (2, 8, 64, 1024):
        bits = B.bit_length() - 1
        for _ in range(200):
            best = rng.integers(0, 4, B).astype(np.float32)     # many ties
            arg = np.arange(B, dtype=np.int32) + 1000
            got = lib.oracle_fps_tree_winner(B, O.P(best), O.P(arg))
            mx = best.max()
            tied = np.nonzero(best == mx)[0]
            rev = [int(format(int(t), f"0{bits}b")[::-1], 2) for t in tied]
            assert got == 1000 + int(tied[int(np.argmin(rev))])


def test_k3_k10_match_reference():
    g = lambda f: K310[f]
    np.testing.assert_array_equal(O.grouping_forward(g("grouping/input"), g("grouping/idx")), g("grouping/output"))
    np.testing.assert_allclose(O.grouping_backward(g("grouping/grad_output"), g("grouping/idx"), g("grouping/input").shape[0]),
                               g("grouping/grad_input"), rtol=0, atol=0)
    np.testing.assert_array_equal(O.interpolation_forward(g("interpolation/input"), g("interpolation/idx"), g("interpolation/weight")),
                                  g("interpolation/output"))
    np.testing.assert_array_equal(O.interpolation_backward(g("interpolation/grad_output"), g("interpolation/idx"), g("interpolation/weight"),
                                                           g("interpolation/input").shape[0]), g("interpolation/grad_input"))
    np.testing.assert_array_equal(O.subtraction_forward(g("subtraction/input1"), g("subtraction/input2"), g("subtraction/idx")),
                                  g("subtraction/output"))
    g1, g2 = O.subtraction_backward(g("subtraction/idx"), g("subtraction/grad_output"))
    np.testing.assert_array_equal(g1, g("subtraction/grad_input1"))
    np.testing.assert_array_equal(g2, g("subtraction/grad_input2"))
    np.testing.assert_array_equal(O.aggregation_forward(g("aggregation/input"), g("aggregation/position"), g("aggregation/weight"), g("aggregation/idx")),
                                  g("aggregation/output"))
    gi, gp, gw = O.aggregation_backward(g("aggregation/input"), g("aggregation/position"), g("aggregation/weight"), g("aggregation/idx"),
                                        g("aggregation/grad_output"))
    np.testing.assert_array_equal(gi, g("aggregation/grad_input"))
    np.testing.assert_array_equal(gp, g("aggregation/grad_position"))
    np.testing.assert_array_equal(gw, g("aggregation/grad_weight"))
