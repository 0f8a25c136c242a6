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


def _tf_sample_case(seed=0, m=60, d=8, ns=9):
    import numpy as np
    rng = np.random.default_rng(seed)
    f = rng.normal(size=(m, d)).astype(np.float32)
    hard = rng.integers(0, 3, m); hard[::11] = -1                     # ignored centres / neighbours
    nbr = np.concatenate([np.arange(m)[:, None], rng.integers(0, m, (m, ns))], 1)
    nbr[rng.random((m, ns + 1)) < 0.15] = m                           # shadow padding of the radius search
    nbr[:, 0] = np.arange(m)
    return rng, f, hard, nbr


def test_tf_sample_strings_and_masks():
    """sample_labels (head.py:551-625): what each segment contributes to the index / positive / negative arrays"""
    import numpy as np
    from oracle import cbl_oracle as C
    rng, f, hard, nbr = _tf_sample_case()
    m, ns = len(f), nbr.shape[1] - 1
    r1, r2 = rng.integers(0, m, (m, 5)), rng.integers(0, m, (m, 4))
    idx, pos, neg = C.tf_samples(hard, nbr, m, "label-nn3-rand5-rand4R", [r1, r2])
    assert idx.shape == (m, ns + 3 + 5 + 4)
    np.testing.assert_array_equal(idx[:, :ns], nbr[:, 1:]); np.testing.assert_array_equal(idx[:, ns:ns + 3], nbr[:, 1:4])
    np.testing.assert_array_equal(idx[:, ns + 3:ns + 8], r1); np.testing.assert_array_equal(idx[:, ns + 8:], r2)
    assert pos[:, ns:ns + 3].all() and not neg[:, ns:ns + 3].any()                       # 'nn': positives, shadow or not
    assert neg[:, ns + 3:ns + 8].all() and not pos[:, ns + 3:].any()                     # 'rand': negatives
    rej = (r2[:, :, None] == nbr[:, None, 1:]).any(-1)
    np.testing.assert_array_equal(neg[:, ns + 8:], ~rej)                                 # 'R': a draw that is a neighbour is not a pair
    assert rej.any()
    lab = np.concatenate([hard, [-1]])[np.minimum(nbr[:, 1:], m)]
    ok = (lab >= 0) & (hard[:, None] >= 0)
    np.testing.assert_array_equal(pos[:, :ns], ok & (lab == hard[:, None])); np.testing.assert_array_equal(neg[:, :ns], ok & (lab != hard[:, None]))
    # the 'label' string alone is the function every earlier test used
    l0 = C.tf_contrast(f, hard, nbr, temperature=0.7)
    l1 = C.tf_contrast(f, hard, nbr, temperature=0.7, sample="label")
    assert l0[0] == l1[0] and np.array_equal(l0[1], l1[1])


def test_tf_sample_and_margin_gradients_agree_with_finite_differences():
    """'nn<k>' / 'rand<n>' / 'rand<n>R' samples (head.py:560-625) and the 'S' margin of both contrasts (:759-760, :783-785): the float32 forward of
    the restatement against its float64 twin, and the stated gradient against central differences of the twin"""
    import numpy as np
    from oracle import cbl_oracle as C
    rng, f, hard, nbr = _tf_sample_case(1)
    m, d = f.shape
    N = len(hard)
    for sample, nr in (("label", ()), ("nn3-rand6", (6,)), ("label-rand5R", (5,)), ("nn2-label-rand4", (4,))):
        rand = [rng.integers(0, m, (m, n)) for n in nr]
        for contrast in ("softnn", "nce"):
            for sep in (False, True):
                kw = dict(temperature=0.7, weight=0.1, contrast=contrast, sample=sample, rand_idx=rand, separate=sep)
                loss, g, mask = C.tf_contrast(f, hard, nbr, **kw)
                assert mask.any() and np.isfinite(loss) and np.isfinite(g).all(), (sample, contrast, sep)
                idx, pos, neg = C.tf_samples(hard, nbr, m, sample, rand)
                rows = np.nonzero(mask)[0]

                def loss64(x):
                    return C.tf_contrast_terms64(x, idx, pos, neg, rows, max(N + 1 - m, 1), 0.7, contrast, sep).mean() * 0.1
                f64 = f.astype(np.float64)
                assert abs(loss64(f64) - loss) <= 1e-5 * abs(loss), (sample, contrast, sep)
                worst = 0.0
                for i in range(0, m, 7):
                    for c in range(0, d, 3):
                        fp, fm = f64.copy(), f64.copy()
                        fp[i, c] += 1e-5; fm[i, c] -= 1e-5
                        num = (loss64(fp) - loss64(fm)) / 2e-5
                        worst = max(worst, abs(num - g[i, c]) / (1e-7 + 1e-4 * max(abs(g[i, c]), np.abs(g).max() * 1e-2)))
                assert worst < 1.0, (sample, contrast, sep, worst)
