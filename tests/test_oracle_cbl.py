"""CPU: oracle/cbl_oracle.py against goldens produced by running the reference's own heads.py / basic_operators.py
(tests/golden/gen_cbl_goldens.py)."""
import os

import numpy as np
import pytest

from oracle import cbl_oracle as C
from tests import oracle_lib as O

G = os.path.join(os.path.dirname(__file__), "golden")
CBL = np.load(os.path.join(G, "cbl_pytorch.npz"))
NSAMPLE = CBL["nsample"]; NSTRIDE = CBL["nstride"]


def stage(case, i, f):
    return CBL[f"{case}/stage{i}/{f}"]


@pytest.mark.parametrize("case", ["default", "temp0p5"])
def test_subscene_labels(case):
    target = CBL[f"{case}/target"]
    p0, o0 = stage(case, 0, "p"), stage(case, 0, "offset")
    np.testing.assert_array_equal(C.one_hot_label(target, 13), stage(case, 0, "soft_label"))
    for i in range(1, 5):
        kr = int(np.prod(NSTRIDE[:i]))                                     # basic_operators.py:22
        idx, _ = O.knnquery(kr, p0, stage(case, i, "p"), o0, stage(case, i, "offset"))
        np.testing.assert_array_equal(C.subscene_label(target, idx, 13), stage(case, i, "soft_label"))


@pytest.mark.parametrize("case", ["default", "temp0p5"])
def test_point_contrast_loss_and_grad(case):
    T = float(CBL[f"{case}/temperature"])
    for i in range(5):
        p, o = stage(case, i, "p"), stage(case, i, "offset")
        idx, _ = O.knnquery(int(NSAMPLE[i]), p, p, o, o)
        loss, grad, mask = C.point_contrast(stage(case, i, "latent"), stage(case, i, "soft_label"), idx, temperature=T, weight=0.1)
        np.testing.assert_allclose(loss, stage(case, i, "loss"), rtol=1e-5)
        np.testing.assert_allclose(grad, stage(case, i, "grad_latent"), rtol=1e-4, atol=1e-7)
        assert mask.any()


def test_boundary_mask():
    g = np.load(os.path.join(G, "boundary_mask.npz"))
    b, p = C.boundary_mask(g["labels"], g["neighbor_label"], get_plain=True)
    np.testing.assert_array_equal(b, g["bound"]); np.testing.assert_array_equal(p, g["plain"])
    np.testing.assert_array_equal(C.boundary_mask(g["labels"], g["neighbor_label"], get_cnt=True), g["cnt"])


def test_boundary_iou_oracle_matches_reference_functions():
    """tool/test.py:392-417: the reference's get_boundary_mask + intersectionAndUnion, run in the build container (gen_cbl_goldens.py)"""
    g = np.load(os.path.join(G, "boundary_mask.npz"))
    r = C.boundary_iou(g["iou_pred"], g["iou_labels"], g["iou_neighbor_idx"], 13, 255)
    for name in ("bound", "plain"):
        for v, key in zip(r[name], "iut"):
            np.testing.assert_array_equal(v, g[f"iou_{name}_{key}"])
    b, p = C.boundary_mask(g["iou_labels"], g["iou_labels"][g["iou_neighbor_idx"]], get_plain=True)
    np.testing.assert_array_equal(b, g["iou_bound_mask"]); np.testing.assert_array_equal(p, g["iou_plain_mask"])


def test_tf_label_kl_known_answers():
    """calc_dist 'kl' (heads/head.py:189-191) as used by sample 'labelkl': hand-computed values, xlogy(0, .) = 0, zero shadow row"""
    import numpy as np
    from oracle import cbl_oracle as C
    p = np.float32([[1, 0, 0], [0.5, 0.5, 0], [0, 0, 1], [0.25, 0.25, 0.5]])
    nb = np.int64([[0, 1, 2, 4], [1, 0, 3, 4], [2, 3, 3, 0], [3, 3, 1, 2]])     # 4 = shadow (N)
    kl = C.tf_label_kl(p, nb)
    big = np.log(np.float32(1.0) / np.float32(1e-12))
    want = np.float32([[0.0, np.log(2.0), big, big],
                       [0.0, 0.5 * np.log(0.5) + 0.5 * np.log(0.5 / 1e-12), 0.5 * np.log(2.0) + 0.5 * np.log(2.0), 0.5 * np.log(0.5 / 1e-12) * 2],
                       [0.0, np.log(2.0), np.log(2.0), big],
                       [0.0, 0.0, 0.25 * np.log(0.5) * 2 + 0.5 * np.log(0.5 / 1e-12), 0.25 * np.log(0.25 / 1e-12) * 2 + 0.5 * np.log(0.5)]])
    np.testing.assert_allclose(kl, want, rtol=1e-5, atol=1e-6)
    # with the threshold: identical distributions are positives, disjoint ones never
    f = np.random.default_rng(0).normal(size=(4, 8)).astype(np.float32)
    loss, g, mask = C.tf_contrast(f, p, np.concatenate([np.arange(4)[:, None], nb], 1), temperature=0.5, weight=0.1, kl_threshold=0.5)
    assert mask.tolist() == [True, True, True, True] and np.isfinite(loss) and np.isfinite(g).all()


def test_nce_restatements_agree_with_finite_differences():
    """contrast 'nce' of both flavours (pytorch heads.py:167-183 — dead code in the reference, see gen_cbl_goldens.py — and TF
    head.py:773-795): the analytic gradients of the restatements against central differences of their own forward"""
    import numpy as np
    from oracle import cbl_oracle as C
    rng = np.random.default_rng(0)
    m, d, ns = 60, 8, 7
    f = rng.normal(size=(m, d)).astype(np.float32)
    hard = rng.integers(0, 3, m)
    nbr = np.concatenate([np.arange(m)[:, None], rng.integers(0, m, (m, ns))], 1)
    for fwd in (lambda x, grad=True: C.point_contrast(x, np.eye(3, dtype=np.float32)[hard], nbr, temperature=0.7, weight=0.1, contrast="nce", grad=grad),
                lambda x, grad=True: C.tf_contrast(x, hard, nbr, temperature=0.7, weight=0.1, contrast="nce", grad=grad)):
        loss, g, mask = fwd(f)
        assert mask.any() and np.isfinite(loss) and loss > 0
        for i in range(0, m, 9):
            for c in range(0, d, 3):
                fp, fm = f.copy(), f.copy()
                fp[i, c] += 1e-2; fm[i, c] -= 1e-2
                num = (fwd(fp, False)[0] - fwd(fm, False)[0]) / 2e-2
                assert abs(num - g[i, c]) < 2e-5 + 2e-2 * abs(g[i, c])
