"""CPU: the Contrastive Boundary Learning head's scatter-flavoured kernels and the label / evaluation passes around it (contrastboundary_amd/csrc/cbl.hip) run from
the host build of the whole library (tests/host_emul/full_library.py, wave semantics) through their C entry points, against the oracle's restatement
(oracle/cbl_oracle.py, pinned by the reference's own heads.py / basic_operators.py through tests/golden/cbl_pytorch.npz and boundary_mask.npz):
  - sub-scene labels (/root/reference/pytorch/model/basic_operators.py:9-50), their arg-max (heads.py:145-149);
  - ContrastHead.point_contrast (heads.py:185-246): two-pass (forward, backward), fused (forward + unit gradient, scaled later), int64 labels;
  - the TF contrast_head (/root/reference/tensorflow/models/heads/head.py:462-807) on shadow-padded radius neighbourhoods with ignored labels, hard labels
    and 'labelkl' soft labels; scene labels through the pools (head.py:25-49, 'max' and 'soft');
  - boundary masks (basic_operators.py:69-97) and the boundary-IoU histogram (tool/test.py:392-417) — against the golden file made by the reference's own code.
The atomic-free pair kernels of the same head are tests/test_cbl_host.py."""
import ctypes
import os

import numpy as np
import pytest

from contrastboundary_amd import synthetic as S
from oracle import cbl_oracle as C
from tests import oracle_lib as O
from tests.host_emul import full_library

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F = ctypes.c_float


@pytest.fixture(scope="module")
def host():
    return full_library.load()


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def knn_table(xyz, k):
    off = np.int32([len(xyz)])
    idx, _ = O.knnquery(k, xyz, xyz, off, off)
    return np.ascontiguousarray(idx, np.int32)


def test_subscene_labels_and_their_argmax(host):
    xyz, lab = S.s_room(2000, seed=1)
    sub = np.ascontiguousarray(xyz[::4])
    idx, _ = O.knnquery(12, xyz, sub, np.int32([2000]), np.int32([len(sub)]))
    idx = np.ascontiguousarray(idx, np.int32)
    m = len(sub)
    out = np.full((m, 13), np.nan, np.float32)
    target = lab.astype(np.int64)
    assert host.cbl_subscene_label(m, 12, 13, P(target), P(idx), P(out), None) == 0
    ref = C.subscene_label(target, idx, 13)
    np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))          # counts / kr: one rounding, the same
    amax = np.full(m, -1, np.int32)
    assert host.cbl_label_argmax(m, 13, P(out), P(amax), None) == 0
    np.testing.assert_array_equal(amax, np.argmax(ref, -1))                           # first maximal index
    assert (np.sort(ref, -1)[:, -1] == np.sort(ref, -1)[:, -2]).any()                 # the case holds tied maxima


@pytest.mark.parametrize("nsample,d,T", [(17, 32, 1.0), (9, 64, 0.5), (37, 16, 2.0), (65, 4, 1.0), (17, 8, 1.0)])
def test_point_contrast_two_pass_and_fused(host, nsample, d, T):
    n = 700
    xyz, lab = S.s_room(n, seed=nsample)
    idx = knn_table(xyz, nsample)
    rng = np.random.default_rng(d)
    f = (rng.normal(size=(n, d)) * 0.5).astype(np.float32)
    amax = lab.astype(np.int32)
    rl, rg, rm = C.point_contrast(f, C.one_hot_label(lab, 13), idx, temperature=T, weight=0.1)
    assert rm.any() and not rm.all()
    gscale = 1e-4 * np.abs(rg).max()

    def outputs():
        return np.full(n, np.nan, np.float32), np.full(n, -1, np.int32), np.full(2, np.nan, np.float32), np.full(1, np.nan, np.float32)
    # two passes
    pp, mask, stats, loss = outputs()
    assert host.cbl_point_contrast_forward(n, nsample, d, P(f), P(amax), P(idx), F(T), F(0.1), P(pp), P(mask), P(stats), P(loss), None) == 0
    np.testing.assert_array_equal(mask > 0, rm)
    assert abs(float(loss[0]) - rl) < 1e-4 * max(1.0, abs(rl)) and int(stats[1]) == int(rm.sum())
    assert (pp[~rm] == 0).all()
    one, g = np.float32([1.0]), np.zeros((n, d), np.float32)
    assert host.cbl_point_contrast_backward(n, nsample, d, P(f), P(amax), P(idx), F(T), F(0.1), P(stats), P(one), P(g), None) == 0
    np.testing.assert_allclose(g, rg, rtol=1e-4, atol=gscale)
    # fused forward + unit gradient, scaled when the upstream gradient arrives (here 0.5)
    pp2, mask2, stats2, loss2 = outputs()
    unit = np.zeros((n, d), np.float32)
    assert host.cbl_point_contrast_forward_grad(n, nsample, d, P(f), P(amax), P(idx), F(T), F(0.1), P(pp2), P(mask2), P(stats2), P(loss2), P(unit), None) == 0
    np.testing.assert_array_equal(mask2, mask)
    assert abs(float(loss2[0]) - float(loss[0])) <= 1e-6 * max(1.0, abs(rl))
    half, g2 = np.float32([0.5]), np.full((n, d), np.nan, np.float32)
    assert host.cbl_contrast_grad_scale(ctypes.c_longlong(n * d), P(unit), P(stats2), P(half), F(0.1), P(g2), None) == 0
    np.testing.assert_allclose(2.0 * g2, rg, rtol=1e-4, atol=gscale)
    # the labels as the reference holds them (torch.long)
    l64 = lab.astype(np.int64)
    pp3, mask3, stats3, loss3 = outputs()
    assert host.cbl_point_contrast_forward_l64(n, nsample, d, P(f), P(l64), P(idx), F(T), F(0.1), P(pp3), P(mask3), P(stats3), P(loss3), None) == 0
    np.testing.assert_array_equal(mask3, mask); np.testing.assert_array_equal(pp3.view(np.uint32), pp.view(np.uint32))
    pp4, mask4, stats4, loss4 = outputs()
    unit4 = np.zeros((n, d), np.float32)
    assert host.cbl_point_contrast_forward_grad_l64(n, nsample, d, P(f), P(l64), P(idx), F(T), F(0.1), P(pp4), P(mask4), P(stats4), P(loss4), P(unit4), None) == 0
    np.testing.assert_array_equal(mask4, mask)
    np.testing.assert_allclose(unit4, unit, rtol=1e-5, atol=1e-5 * np.abs(unit).max())      # float atomics: summation order only


def test_no_boundary_point_gives_zero_loss_and_zero_gradient(host):
    n, d = 300, 16
    xyz, _ = S.s_room(n, seed=2)
    idx = knn_table(xyz, 9)
    f = np.random.default_rng(0).normal(size=(n, d)).astype(np.float32)
    amax = np.full(n, 3, np.int32)                                                  # one class everywhere: no point has a negative neighbour (heads.py:233)
    pp, mask, stats, loss = np.full(n, np.nan, np.float32), np.full(n, -1, np.int32), np.full(2, np.nan, np.float32), np.full(1, np.nan, np.float32)
    unit = np.zeros((n, d), np.float32)
    assert host.cbl_point_contrast_forward_grad(n, 9, d, P(f), P(amax), P(idx), F(1.0), F(0.1), P(pp), P(mask), P(stats), P(loss), P(unit), None) == 0
    assert float(loss[0]) == 0.0 and not mask.any() and float(stats[1]) == 0.0
    g, one = np.full((n, d), np.nan, np.float32), np.float32([1.0])
    assert host.cbl_contrast_grad_scale(ctypes.c_longlong(n * d), P(unit), P(stats), P(one), F(0.1), P(g), None) == 0
    assert (g == 0).all()


def radius_scene(n, lens, r, limit, seed, ignored=0):
    xyz, lab = S.s_room(n, seed=seed)
    lab = lab.copy()
    if ignored:
        lab[np.random.default_rng(seed).choice(n, ignored, replace=False)] = -1
    lens = np.int32(lens)
    nb, _, _ = O.radius_neighbors(xyz, xyz, lens, lens, r, limit)
    return xyz, lab, np.ascontiguousarray(nb, np.int32)


@pytest.mark.parametrize("limit,d,T", [(26, 32, 1.0), (41, 16, 0.5), (12, 64, 2.0)])
def test_tf_contrast_head_with_shadow_neighbours_and_ignored_labels(host, limit, d, T):
    n = 1500
    xyz, lab, nb = radius_scene(n, [600, 900], 0.2, limit, seed=limit, ignored=80)
    assert (nb == n).any()                                                          # shadow entries present
    f = (np.random.default_rng(limit).normal(size=(n, d)) * 0.5).astype(np.float32)
    lab32 = lab.astype(np.int32)
    rl, rg, rm = C.tf_contrast(f, lab, nb, temperature=T, weight=0.1)
    assert rm.any() and not rm.all()
    pp, mask, stats, loss = np.full(n, np.nan, np.float32), np.full(n, -1, np.int32), np.full(2, np.nan, np.float32), np.full(1, np.nan, np.float32)
    assert host.cbl_tf_contrast_forward(n, n, limit, d, P(f), P(lab32), P(nb), F(T), F(0.1), P(pp), P(mask), P(stats), P(loss), None) == 0
    np.testing.assert_array_equal(mask > 0, rm)
    assert abs(float(loss[0]) - rl) < 1e-4 * max(1.0, abs(rl))
    one, g = np.float32([1.0]), np.zeros((n, d), np.float32)
    assert host.cbl_tf_contrast_backward(n, n, limit, d, P(f), P(lab32), P(nb), F(T), F(0.1), P(stats), P(one), P(g), None) == 0
    np.testing.assert_allclose(g, rg, rtol=1e-4, atol=1e-4 * np.abs(rg).max())
    pp2, mask2, stats2, loss2 = np.full(n, np.nan, np.float32), np.full(n, -1, np.int32), np.full(2, np.nan, np.float32), np.full(1, np.nan, np.float32)
    unit, g2 = np.zeros((n, d), np.float32), np.full((n, d), np.nan, np.float32)
    assert host.cbl_tf_contrast_forward_grad(n, n, limit, d, P(f), P(lab32), P(nb), F(T), F(0.1), P(pp2), P(mask2), P(stats2), P(loss2), P(unit), None) == 0
    assert host.cbl_contrast_grad_scale(ctypes.c_longlong(n * d), P(unit), P(stats2), P(one), F(0.1), P(g2), None) == 0
    np.testing.assert_array_equal(mask2, mask)
    np.testing.assert_allclose(g2, rg, rtol=1e-4, atol=1e-4 * np.abs(rg).max())


def test_scene_labels_through_the_pools_and_the_labelkl_head(host):
    """head.py:25-49 ('max' as a histogram whose arg-max is taken, 'soft' by the number of valid neighbours) on a sub-sampled layer, then sample 'labelkl<thr>'
    (head.py:492-519) on that layer's radius neighbourhoods with the soft labels"""
    n = 2600
    xyz, lab = S.s_room(n, seed=9)
    lab = lab.copy(); lab[::37] = -1
    lens = np.int32([1100, 1500])
    sub, sl = O.grid_subsampling(xyz, lens, 0.16)
    sub, sl = np.ascontiguousarray(sub), sl.astype(np.int32)
    m = len(sub)
    pools, _, _ = O.radius_neighbors(sub, xyz, sl, lens, 0.16, 32)
    pools = np.ascontiguousarray(pools, np.int32)
    l64 = lab.astype(np.int64)
    hist, soft = np.full((m, 13), np.nan, np.float32), np.full((m, 13), np.nan, np.float32)
    assert host.cbl_tf_scene_label(m, n, 32, 13, P(l64), P(pools), 0, P(hist), None) == 0
    assert host.cbl_tf_scene_label(m, n, 32, 13, P(l64), P(pools), 1, P(soft), None) == 0
    hard = np.full(m, -1, np.int32)
    assert host.cbl_label_argmax(m, 13, P(hist), P(hard), None) == 0
    np.testing.assert_array_equal(hard, C.tf_scene_label(lab, pools, 13, "max"))
    ref_soft = C.tf_scene_label(lab, pools, 13, "soft")
    np.testing.assert_allclose(soft, ref_soft, rtol=1e-6, atol=1e-7)
    limit, d, T, thr = 20, 32, 0.5, 0.5
    nb, _, _ = O.radius_neighbors(sub, sub, sl, sl, 0.4, limit)
    nb = np.ascontiguousarray(nb, np.int32)
    assert (nb == m).any()
    kl = C.tf_label_kl(soft, nb[:, 1:])
    assert np.abs(kl - thr).min() > 1e-4                                            # no pair on the threshold: logf rounding cannot flip one
    f = (np.random.default_rng(3).normal(size=(m, d)) * 0.5).astype(np.float32)
    rl, rg, rm = C.tf_contrast(f, soft, nb, temperature=T, weight=0.1, kl_threshold=thr)
    assert rm.any() and not rm.all()
    pp, mask, stats, loss = np.full(m, np.nan, np.float32), np.full(m, -1, np.int32), np.full(2, np.nan, np.float32), np.full(1, np.nan, np.float32)
    assert host.cbl_tf_contrast_forward_kl(m, m, limit, d, P(f), P(soft), 13, F(thr), P(nb), F(T), F(0.1), P(pp), P(mask), P(stats), P(loss), None) == 0
    np.testing.assert_array_equal(mask > 0, rm)
    assert abs(float(loss[0]) - rl) < 1e-4 * max(1.0, abs(rl))
    unit, g, one = np.zeros((m, d), np.float32), np.full((m, d), np.nan, np.float32), np.float32([1.0])
    assert host.cbl_tf_contrast_forward_grad_kl(m, m, limit, d, P(f), P(soft), 13, F(thr), P(nb), F(T), F(0.1), P(pp), P(mask), P(stats), P(loss), P(unit), None) == 0
    assert host.cbl_contrast_grad_scale(ctypes.c_longlong(m * d), P(unit), P(stats), P(one), F(0.1), P(g), None) == 0
    np.testing.assert_allclose(g, rg, rtol=1e-4, atol=1e-4 * np.abs(rg).max())


def test_boundary_masks_and_the_boundary_iou_histogram_against_the_reference_goldens(host):
    g = np.load(os.path.join(G, "boundary_mask.npz"))
    pred, labels, idx = g["iou_pred"].astype(np.int64), g["iou_labels"].astype(np.int64), np.ascontiguousarray(g["iou_neighbor_idx"], np.int32)
    n, k = idx.shape
    hist = np.zeros((2, 3, 13), np.uint64)
    assert host.cbl_boundary_iou(n, k, 13, ctypes.c_longlong(255), P(pred), P(labels), P(idx), P(hist), None) == 0
    for a, name in enumerate(("bound", "plain")):
        i, o, t = hist[a].astype(np.int64)
        np.testing.assert_array_equal(i, g["iou_%s_i" % name])
        np.testing.assert_array_equal(o + t - i, g["iou_%s_u" % name])             # union = output + target - intersection (common_util.py:36)
        np.testing.assert_array_equal(t, g["iou_%s_t" % name])
    # the masks themselves, with invalid (negative) neighbour labels in play
    lab = labels.copy(); lab[::11] = -1
    bound, plain, cnt = np.full(n, 7, np.uint8), np.full(n, 7, np.uint8), np.full(n, -1, np.int32)
    assert host.cbl_boundary_mask(n, k, P(lab), P(idx), P(bound), P(plain), P(cnt), None) == 0
    rb, rp = C.boundary_mask(lab, lab[idx], get_plain=True)
    np.testing.assert_array_equal(bound.astype(bool), rb); np.testing.assert_array_equal(plain.astype(bool), rp)
    np.testing.assert_array_equal(cnt, C.boundary_mask(lab, lab[idx], get_cnt=True))
    assert host.cbl_boundary_mask(n, k, P(lab), P(idx), None, None, P(cnt), None) == 0    # any output may be NULL
