"""GPU parity: fused CBL head (csrc/cbl.hip through the C ABI) vs goldens from the reference's own heads.py and vs the oracle."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import cbl_oracle as C
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
CBL = np.load(os.path.join(G, "cbl_pytorch.npz"))
TOL = 1e-4        # north_star: float outputs within 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


class Cfg(dict):
    __getattr__ = dict.__getitem__


def make_stage_list(case):
    up = []
    for i in range(5):
        up.append({"p_out": dev(CBL[f"{case}/stage{i}/p"]), "offset": dev(CBL[f"{case}/stage{i}/offset"]),
                   "latent": dev(CBL[f"{case}/stage{i}/latent"]).requires_grad_(True), "f_out": None})
    return {"inputs": None, "up": up, "down": up}


@pytest.mark.parametrize("case", ["default", "temp0p5"])
def test_contrast_head_matches_reference(case):
    from contrastboundary_amd.heads import ContrastHead
    cfg = Cfg(nsample=[36, 24, 24, 24, 24], nstride=[4, 4, 4, 4], num_classes=13, num_layers=5, voxel_size=0.04,
              contrast=Cfg(stage="Ua", contrast="softnn", ftype="latent", sample="label", pos="cnt", dist="l2",
                           temperature=float(CBL[f"{case}/temperature"]), weight="w.1"))
    head = ContrastHead(cfg.contrast, cfg)
    sl = make_stage_list(case)
    losses = head(None, dev(CBL[f"{case}/target"]), sl)
    assert len(losses) == 5
    torch.stack(losses).sum().backward()
    for i in range(5):
        np.testing.assert_allclose(losses[i].item(), CBL[f"{case}/stage{i}/loss"], rtol=TOL, atol=1e-6)
        np.testing.assert_allclose(sl["up"][i]["latent"].grad.cpu().numpy(), CBL[f"{case}/stage{i}/grad_latent"], rtol=1e-3, atol=TOL * 1e-2)


@pytest.mark.parametrize("case", ["default", "temp0p5"])
def test_subscene_labels_match_reference(case):
    from contrastboundary_amd.basic_operators import get_subscene_label
    sl = make_stage_list(case)
    target = dev(CBL[f"{case}/target"])
    for i in range(5):
        soft = get_subscene_label("up", i, sl, target, [4, 4, 4, 4], 13)
        np.testing.assert_array_equal(soft.cpu().numpy(), CBL[f"{case}/stage{i}/soft_label"])       # counts / kr: exact


@pytest.mark.parametrize("nsample,d", [(8, 16), (17, 32), (33, 32), (40, 64), (65, 8)])
def test_point_contrast_vs_oracle(nsample, d):
    """every group width (16/32/64 lanes) and feature width, random blocky labels"""
    from contrastboundary_amd import heads, pointops
    rng = np.random.default_rng(nsample)
    n = 3000
    xyz = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    lab = (np.floor(xyz[:, 0] * 4) + 4 * np.floor(xyz[:, 1] * 3)).astype(np.int64) % 13
    feat = (rng.normal(size=(n, d)) * 0.5).astype(np.float32)
    off = np.int32([1200, 3000])
    idx, _ = O.knnquery(nsample, xyz, xyz, off, off)
    f = dev(feat).requires_grad_(True)
    loss, mask = heads.point_contrast(f, dev(lab), dev(idx), temperature=0.7, weight=0.1, return_mask=True)
    loss.backward()
    rloss, rgrad, rmask = C.point_contrast(feat, np.eye(13, dtype=np.float32)[lab], idx, temperature=0.7, weight=0.1)
    np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), rmask)
    np.testing.assert_allclose(loss.item(), rloss, rtol=TOL)
    np.testing.assert_allclose(f.grad.cpu().numpy(), rgrad, rtol=1e-4, atol=1e-4 * np.abs(rgrad).max())


def test_no_boundary_point_gives_zero_loss_and_zero_grad():
    from contrastboundary_amd import heads
    rng = np.random.default_rng(0)
    n = 500
    xyz = rng.uniform(size=(n, 3)).astype(np.float32)
    idx, _ = O.knnquery(8, xyz, xyz, [n], [n])
    f = dev(rng.normal(size=(n, 32)).astype(np.float32)).requires_grad_(True)
    loss = heads.point_contrast(f, dev(np.zeros(n, np.int64)), dev(idx))      # a single class: no negatives anywhere (heads.py:233)
    loss.backward()
    assert loss.item() == 0.0 and float(f.grad.abs().max()) == 0.0


def test_boundary_mask():
    from contrastboundary_amd.basic_operators import get_boundary_mask
    g = np.load(os.path.join(G, "boundary_mask.npz"))
    b, p = get_boundary_mask(dev(g["labels"]), neighbor_label=dev(g["neighbor_label"]), get_plain=True)
    np.testing.assert_array_equal(b.cpu().numpy(), g["bound"]); np.testing.assert_array_equal(p.cpu().numpy(), g["plain"])
    # native path from indices, labels with invalid (-1) entries
    rng = np.random.default_rng(1)
    lab = rng.integers(-1, 5, 4000).astype(np.int64); nidx = rng.integers(0, 4000, (4000, 16)).astype(np.int32)
    b, p = get_boundary_mask(dev(lab), neighbor_idx=dev(nidx), get_plain=True)
    c = get_boundary_mask(dev(lab), neighbor_idx=dev(nidx), get_cnt=True)
    rb, rp = C.boundary_mask(lab, lab[nidx], get_plain=True)
    np.testing.assert_array_equal(b.cpu().numpy(), rb); np.testing.assert_array_equal(p.cpu().numpy(), rp)
    np.testing.assert_array_equal(c.cpu().numpy(), C.boundary_mask(lab, lab[nidx], get_cnt=True))


def test_cbl_full_size_properties():
    """N=40960, nsample=36 (stage 0 of the shipped config): loss invariant to a global feature translation, scales with weight,
    gradient sums to zero over all points (every pair contributes +c to one point and -c to another)."""
    from contrastboundary_amd import heads, hotpath, pointops
    sc = hotpath.Scene.synthetic(40960, 32, seed=0)
    idx, _ = pointops.knnquery_raw(36, sc.xyz, sc.xyz, sc.offset, sc.offset)
    f = sc.feat.clone().requires_grad_(True)
    l1 = heads.point_contrast(f, sc.labels, idx, 1.0, 0.1)
    l1.backward()
    l2 = heads.point_contrast(sc.feat + 3.0, sc.labels, idx, 1.0, 0.1)
    l3 = heads.point_contrast(sc.feat, sc.labels, idx, 1.0, 0.3)
    assert abs(l1.item() - l2.item()) < 1e-4 and abs(3 * l1.item() - l3.item()) < 1e-4 and l1.item() > 0
    assert float(f.grad.sum(0).abs().max()) < 1e-4


def test_hard_labels_as_int64_and_int32_give_the_same_loss_and_gradient():
    """torch.long targets go to the kernels as they are (cbl_point_contrast_forward*_l64); an int32 copy takes the original entry points"""
    from contrastboundary_amd import heads, hotpath, pointops
    sc = hotpath.Scene.synthetic(8192, 32, seed=4)
    idx, _ = pointops.knnquery_raw(24, sc.xyz, sc.xyz, sc.offset, sc.offset)
    lab64 = sc.labels.clone(); lab64[::7] = -100                               # an ignore label: compared for equality like any other id
    res = []
    for lab in (lab64, lab64.to(torch.int32)):
        f = sc.latent.clone().requires_grad_(True)
        loss = heads.point_contrast(f, lab, idx, 0.7, 0.1)
        loss.backward()
        with torch.no_grad():
            res.append((loss.detach().clone(), f.grad.clone(), heads.point_contrast(sc.latent, lab, idx, 0.7, 0.1)))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2]) and res[0][0].item() > 0
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-5, atol=1e-8)       # atomics: summation order only


@pytest.mark.parametrize("atomic", [False, True])
@pytest.mark.parametrize("limit,d,T", [(26, 32, 1.0), (41, 16, 0.5), (12, 64, 2.0)])
def test_tf_contrast_head_vs_oracle(limit, d, T, atomic):
    """a16: TF contrast_head on radius neighbourhoods with shadow padding + ignored (-1) labels, vs the numpy restatement; the default route (pair
    kernels, gradient as a gather) and round 1's kernels (`atomic_scatter`)"""
    from contrastboundary_amd import heads, tf_ops
    from contrastboundary_amd import synthetic as S
    xyz, lab = S.s_room(6000, seed=limit)
    rng = np.random.default_rng(limit)
    lab = lab.copy(); lab[rng.choice(6000, 300, replace=False)] = -1          # ignored points
    lens = np.int32([2500, 3500])
    nb = tf_ops.tf_batch_neighbors(dev(xyz), dev(xyz), dev(lens), dev(lens), 0.12, limit, exact_shape=False)
    feat = (rng.normal(size=(6000, d)) * 0.5).astype(np.float32)
    f = dev(feat).requires_grad_(True)
    loss, mask = heads.tf_contrast(f, dev(lab), nb, T, 0.1, return_mask=True, atomic_scatter=atomic)
    loss.backward()
    rl, rg, rm = C.tf_contrast(feat, lab, nb.cpu().numpy(), temperature=T, weight=0.1)
    np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), rm)
    np.testing.assert_allclose(loss.item(), rl, rtol=TOL)
    np.testing.assert_allclose(f.grad.cpu().numpy(), rg, rtol=1e-4, atol=1e-4 * np.abs(rg).max())
    assert (nb.cpu().numpy() == 6000).any()                                   # the case really contains shadow entries


@pytest.mark.parametrize("atomic", [False, True])
@pytest.mark.parametrize("limit,d,T,thr", [(26, 32, 0.5, 0.5), (20, 64, 1.0, 0.3)])
def test_tf_contrast_head_labelkl_vs_oracle(limit, d, T, thr, atomic):
    """sample 'labelkl<thr>' (s3dis.py:162-163): positives by the KL divergence of the sub-scene label distributions, on a sub-sampled
    stage (soft labels from the stage-0 points) with shadow-padded radius neighbourhoods"""
    from contrastboundary_amd import heads, tf_ops
    from contrastboundary_amd import synthetic as S
    xyz, lab = S.s_room(9000, seed=limit + 1)
    lens = np.int32([4000, 5000])
    sub, sl = tf_ops.tf_batch_subsampling(dev(xyz), dev(lens), 0.10)
    scene_nb = tf_ops.tf_batch_neighbors(sub.contiguous(), dev(xyz), sl, dev(lens), 0.10, 32, exact_shape=False)
    soft = heads.tf_scene_label(dev(lab), scene_nb, 13, "soft")                 # (m, 13) distributions with small denominators
    m = sub.shape[0]
    nb = tf_ops.tf_batch_neighbors(sub.contiguous(), sub.contiguous(), sl, sl, 0.25, limit, exact_shape=False)
    rng = np.random.default_rng(limit)
    feat = (rng.normal(size=(m, d)) * 0.5).astype(np.float32)
    f = dev(feat).requires_grad_(True)
    loss, mask = heads.tf_contrast(f, soft, nb, T, 0.1, return_mask=True, kl_threshold=thr, atomic_scatter=atomic)
    loss.backward()
    soft_h, nb_h = soft.cpu().numpy(), nb.cpu().numpy()
    kl = C.tf_label_kl(soft_h, nb_h[:, 1:])
    assert np.abs(kl - thr).min() > 1e-4                                       # no pair sits on the threshold: logf rounding cannot flip one
    rl, rg, rm = C.tf_contrast(feat, soft_h, nb_h, temperature=T, weight=0.1, kl_threshold=thr)
    np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), rm)
    assert rm.any() and not rm.all()
    np.testing.assert_allclose(loss.item(), rl, rtol=TOL)
    np.testing.assert_allclose(f.grad.cpu().numpy(), rg, rtol=1e-4, atol=1e-4 * np.abs(rg).max())
    assert (nb_h == m).any()                                                   # shadow entries present
    # inference path (no gradient): same loss
    with torch.no_grad():
        l2 = heads.tf_contrast(dev(feat), soft, nb, T, 0.1, kl_threshold=thr, atomic_scatter=atomic)
    np.testing.assert_allclose(l2.item(), rl, rtol=TOL)


def test_tf_scene_labels_vs_oracle():
    from contrastboundary_amd import heads, tf_ops
    from contrastboundary_amd import synthetic as S
    xyz, lab = S.s_room(8000, seed=9)
    lab = lab.copy(); lab[::37] = -1
    lens = np.int32([8000])
    sub, sl = tf_ops.tf_batch_subsampling(dev(xyz), dev(lens), 0.16)
    nb = tf_ops.tf_batch_neighbors(sub.contiguous(), dev(xyz), sl, dev(lens), 0.16, 48, exact_shape=False)
    hard = heads.tf_scene_label(dev(lab), nb, 13, "max")
    soft = heads.tf_scene_label(dev(lab), nb, 13, "soft")
    np.testing.assert_array_equal(hard.cpu().numpy(), C.tf_scene_label(lab, nb.cpu().numpy(), 13, "max"))
    np.testing.assert_allclose(soft.cpu().numpy(), C.tf_scene_label(lab, nb.cpu().numpy(), 13, "soft"), rtol=1e-6, atol=1e-7)


def test_boundary_iou_evaluation():
    """(f) rank 3: kr-neighbourhood search + boundary / plain masks + masked intersection-and-union, one call per room"""
    from contrastboundary_amd.basic_operators import boundary_iou
    g = np.load(os.path.join(G, "boundary_mask.npz"))
    # a) from the golden's neighbour indices: the reference's own numbers
    r = boundary_iou(dev(g["iou_pred"]), dev(g["iou_labels"]), neighbor_idx=dev(g["iou_neighbor_idx"]), num_classes=13, ignore_label=255)
    for name in ("bound", "plain"):
        for v, key in zip(r[name], "iut"):
            np.testing.assert_array_equal(v.cpu().numpy(), g[f"iou_{name}_{key}"])
    # b) end to end from coordinates (the search is part of the call)
    n = len(g["iou_labels"])
    r2 = boundary_iou(dev(g["iou_pred"]), dev(g["iou_labels"]), xyz=dev(g["iou_xyz"]), offset=dev(np.int32([n])), kr=8, num_classes=13, ignore_label=255)
    for name in ("bound", "plain"):
        for a, b in zip(r[name], r2[name]):
            assert torch.equal(a, b)
    # c) room-sized property check: every non-ignored point is counted once in 'bound' or 'plain' targets... or in neither when it has
    #    both no differing and (impossible) — i.e. bound and plain partition the points whose neighbour labels are all valid
    from contrastboundary_amd import synthetic as S
    xyz, lab = S.s_room(200000, seed=3, scale=4.0)
    pred = lab.copy(); pred[::5] = (pred[::5] + 1) % 13
    r3 = boundary_iou(dev(pred), dev(lab), xyz=dev(xyz), offset=dev(np.int32([200000])), kr=16, num_classes=13)
    t_total = (r3["bound"][2] + r3["plain"][2]).cpu().numpy()
    np.testing.assert_array_equal(t_total, np.bincount(lab, minlength=13))
    ref = C.boundary_iou(pred[:0], lab[:0], np.zeros((0, 16), np.int64), 13)
    assert all(int(v.sum()) == 0 for v in ref["bound"])


@pytest.mark.parametrize("tf_variant", [False, True])
def test_fused_forward_gradient_equals_the_two_pass_entries(tf_variant):
    """cbl_*_contrast_forward_grad + cbl_contrast_grad_scale (training path of the mirrors) against cbl_*_contrast_forward / _backward"""
    import ctypes
    from contrastboundary_amd import _lib, pointops
    L = _lib.lib()
    rng = np.random.default_rng(4)
    n, d, K = 3000, 32, 25
    xyz = dev(rng.uniform(0, 1, (n, 3)).astype(np.float32)); o = dev(np.int32([n]))
    feat = dev(rng.normal(size=(n, d)).astype(np.float32))
    lab = dev(rng.integers(0, 5, n).astype(np.int32))
    nidx, _ = pointops.knnquery_raw(K, xyz, xyz, o, o, algo="set")
    st = _lib.stream_of(feat)
    ci, cf = ctypes.c_int, ctypes.c_float
    mk = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device="cuda")
    pp1, m1, s1, l1 = mk(n), mk(n, dt=torch.int32), mk(2), mk(1)
    pp2, m2, s2, l2 = mk(n), mk(n, dt=torch.int32), mk(2), mk(1)
    unit = torch.zeros(n, d, device="cuda"); g1 = torch.zeros(n, d, device="cuda"); g2 = mk(n, d)
    gl = dev(np.float32([0.7]))
    if tf_variant:
        pre = (ci(n), ci(n), ci(K), ci(d), _lib.ptr(feat), _lib.ptr(lab), _lib.ptr(nidx), cf(0.5), cf(0.1))
        _lib.check(L.cbl_tf_contrast_forward(*pre, _lib.ptr(pp1), _lib.ptr(m1), _lib.ptr(s1), _lib.ptr(l1), st), "fwd")
        _lib.check(L.cbl_tf_contrast_backward(*pre, _lib.ptr(s1), _lib.ptr(gl), _lib.ptr(g1), st), "bwd")
        _lib.check(L.cbl_tf_contrast_forward_grad(*pre, _lib.ptr(pp2), _lib.ptr(m2), _lib.ptr(s2), _lib.ptr(l2), _lib.ptr(unit), st), "fwd_grad")
    else:
        pre = (ci(n), ci(K), ci(d), _lib.ptr(feat), _lib.ptr(lab), _lib.ptr(nidx), cf(0.5), cf(0.1))
        _lib.check(L.cbl_point_contrast_forward(*pre, _lib.ptr(pp1), _lib.ptr(m1), _lib.ptr(s1), _lib.ptr(l1), st), "fwd")
        _lib.check(L.cbl_point_contrast_backward(*pre, _lib.ptr(s1), _lib.ptr(gl), _lib.ptr(g1), st), "bwd")
        _lib.check(L.cbl_point_contrast_forward_grad(*pre, _lib.ptr(pp2), _lib.ptr(m2), _lib.ptr(s2), _lib.ptr(l2), _lib.ptr(unit), st), "fwd_grad")
    _lib.check(L.cbl_contrast_grad_scale(ctypes.c_longlong(n * d), _lib.ptr(unit), _lib.ptr(s2), _lib.ptr(gl), cf(0.1), _lib.ptr(g2), st), "scale")
    assert torch.equal(pp1, pp2) and torch.equal(m1, m2) and torch.equal(l1, l2) and float(s1[1]) > 100
    np.testing.assert_allclose(g2.cpu().numpy(), g1.cpu().numpy(), rtol=1e-4, atol=1e-6 * float(g1.abs().max()))


def test_contrast_head_with_projection_matches_reference():
    """head_cfg.project (heads.py:88-92, 187-188): an MLPbyOps in front of the contrast, on the stages' own widths (32 ... 512 channels);
    the reference's seeded parameters are loaded into the mirror (same module names), BatchNorm in train mode as the criterion runs"""
    from contrastboundary_amd.heads import ContrastHead
    case = "project"
    cfg = Cfg(nsample=[36, 24, 24, 24, 24], nstride=[4, 4, 4, 4], num_classes=13, num_layers=5, voxel_size=0.04, base_fdim=32,
              contrast=Cfg(stage="Ua", contrast="softnn", ftype="f_out", sample="label", pos="cnt", dist="l2", project="mlp2",
                           temperature=float(CBL[f"{case}/temperature"]), weight="w.1"))
    head = ContrastHead(cfg.contrast, cfg).cuda()
    state = {k[len(f"{case}/state/"):]: torch.from_numpy(CBL[k]) for k in CBL.files if k.startswith(f"{case}/state/")}
    assert set(state) == set(head.state_dict()), "module / parameter names differ from the reference's"
    head.load_state_dict(state)
    head.train()
    up = [{"p_out": dev(CBL[f"{case}/stage{i}/p"]), "offset": dev(CBL[f"{case}/stage{i}/offset"]), "latent": None,
           "f_out": dev(CBL[f"{case}/stage{i}/f_out"]).requires_grad_(True)} for i in range(5)]
    sl = {"inputs": None, "up": up, "down": up}
    losses = head(None, dev(CBL[f"{case}/target"]), sl)
    torch.stack(losses).sum().backward()
    for i in range(5):
        np.testing.assert_allclose(losses[i].item(), CBL[f"{case}/stage{i}/loss"], rtol=1e-3, atol=1e-6)      # through Linear + train-mode BatchNorm + Linear
        ref = CBL[f"{case}/stage{i}/grad_f_out"]
        np.testing.assert_allclose(up[i]["f_out"].grad.cpu().numpy(), ref, rtol=2e-2, atol=2e-3 * np.abs(ref).max())


@pytest.mark.parametrize("nsample,d,temperature", [(8, 16, 0.7), (24, 32, 1.0), (36, 32, 0.5), (40, 64, 1.3)])
def test_point_contrast_nce_vs_oracle(nsample, d, temperature):
    """contrast 'nce' (heads.py:167-183) — dead code in the reference itself (`1 - posmask` on a bool mask raises), so pinned by the
    restatement only: one -log(e_j / (e_j + negatives)) per positive pair, mean over all positives"""
    from contrastboundary_amd import heads
    rng = np.random.default_rng(nsample + 100)
    n = 3000
    xyz = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    lab = (np.floor(xyz[:, 0] * 4) + 4 * np.floor(xyz[:, 1] * 3)).astype(np.int64) % 13
    feat = (rng.normal(size=(n, d)) * 0.5).astype(np.float32)
    off = np.int32([1200, 3000])
    idx, _ = O.knnquery(nsample, xyz, xyz, off, off)
    f = dev(feat).requires_grad_(True)
    loss, mask = heads.point_contrast(f, dev(lab), dev(idx), temperature=temperature, weight=0.1, return_mask=True, contrast="nce")
    loss.backward()
    rloss, rgrad, rmask = C.point_contrast(feat, np.eye(13, dtype=np.float32)[lab], idx, temperature=temperature, weight=0.1, contrast="nce")
    np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), rmask)
    np.testing.assert_allclose(loss.item(), rloss, rtol=TOL)
    np.testing.assert_allclose(f.grad.cpu().numpy(), rgrad, rtol=1e-4, atol=1e-4 * np.abs(rgrad).max())


@pytest.mark.parametrize("k,d", [(16, 32), (27, 16)])
def test_tf_contrast_nce_vs_oracle(k, d):
    """TF contrast_head with contrast 'nce' (tensorflow/models/heads/head.py:773-795, no 'S' margin, no masking) on radius neighbourhoods
    with shadow padding and ignored labels; parity unpinned by execution (TensorFlow absent): against the restatement"""
    from contrastboundary_amd import heads
    rng = np.random.default_rng(k)
    n = 2500
    xyz = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    lab = (np.floor(xyz[:, 0] * 3) + 3 * np.floor(xyz[:, 2] * 3)).astype(np.int64) % 7
    lab[::53] = -1                                                        # ignored points
    feat = (rng.normal(size=(n, d)) * 0.5).astype(np.float32)
    idx, _ = O.knnquery(k, xyz, xyz, np.int32([n]), np.int32([n]))
    nb = idx.copy()
    npad = rng.integers(0, k // 3, n)
    for i in range(n):
        if npad[i]:
            nb[i, k - npad[i]:] = n                                       # the radius search's shadow index
    f = dev(feat).requires_grad_(True)
    loss, mask = heads.tf_contrast(f, dev(lab), dev(nb.astype(np.int32)), temperature=0.8, weight=0.1, return_mask=True, contrast="nce")
    loss.backward()
    rloss, rgrad, rmask = C.tf_contrast(feat, lab, nb, temperature=0.8, weight=0.1, contrast="nce")
    np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), rmask)
    np.testing.assert_allclose(loss.item(), rloss, rtol=TOL)
    np.testing.assert_allclose(f.grad.cpu().numpy(), rgrad, rtol=1e-4, atol=1e-4 * np.abs(rgrad).max())


def _tf_radius_case(seed, n, k, d):
    """radius-search-shaped neighbourhoods: kNN rows with a random number of trailing shadow entries (index n), some ignored labels"""
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    lab = (np.floor(xyz[:, 0] * 3) + 3 * np.floor(xyz[:, 2] * 3)).astype(np.int64) % 7
    lab[::53] = -1                                                        # ignored points
    feat = (rng.normal(size=(n, d)) * 0.5).astype(np.float32)
    lens = np.int32([n // 3, n - n // 3])
    off = np.cumsum(lens).astype(np.int32)
    idx, _ = O.knnquery(k, xyz, xyz, off, off)
    nb = idx.copy()
    npad = rng.integers(0, k // 3, n)
    for i in range(n):
        if npad[i]:
            nb[i, k - npad[i]:] = n                                       # the radius search's shadow index
    nb[::97, 2:] = n                                                      # sparse spots: one real neighbour, the rest shadow
    return rng, feat, lab, nb.astype(np.int32), lens


@pytest.mark.parametrize("sample,nr", [("label", ()), ("nn4-rand12", (12,)), ("label-rand8R", (8,)), ("nn2-label-rand6", (6,)), ("label-nn3-rand5-rand4R", (5, 4))])
@pytest.mark.parametrize("contrast", ["softnn", "nce"])
@pytest.mark.parametrize("separate", [False, True])
def test_tf_contrast_samples_and_margin_vs_oracle(sample, nr, contrast, separate):
    """TF contrast_head's sample strings beyond 'label' ('nn<k>', 'rand<n>', 'rand<n>R', head.py:560-625) and the 'S' margin of both contrasts
    (:759-760, :783-785), with shadow padding (an 'nn' column may be a shadow: a positive at the zero row) and ignored labels; the random draws are
    handed in (tf.random.uniform cannot be replayed).  Parity unpinned by execution (TensorFlow absent): against the restatement, whose gradient
    tests/test_oracle_cbl.py checks against finite differences."""
    from contrastboundary_amd import heads
    n, k, d = 2400, 14, 32
    rng, feat, lab, nb, lens = _tf_radius_case(len(sample) + 7 * len(nr), n, k, d)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    cloud = np.repeat(np.arange(len(lens)), lens)
    rand = []
    for r in nr:
        draw = (rng.random((n, r)) * lens[cloud][:, None]).astype(np.int64) + starts[cloud][:, None]
        draw[:, 0] = np.where(rng.random(n) < 0.3, nb[:, 1], draw[:, 0])   # some draws ARE neighbours (what 'R' rejects)
        draw = np.minimum(draw, n - 1)
        rand.append(draw.astype(np.int32))
    f = dev(feat).requires_grad_(True)
    loss, mask = heads.tf_contrast(f, dev(lab), dev(nb), temperature=0.8, weight=0.1, return_mask=True, contrast=contrast, sample=sample,
                                   margin="S" if separate else None, rand_idx=[dev(r) for r in rand])
    loss.backward()
    rloss, rgrad, rmask = C.tf_contrast(feat, lab, nb, temperature=0.8, weight=0.1, contrast=contrast, sample=sample, rand_idx=rand, separate=separate)
    assert rmask.any() and not rmask.all() or "nn" in sample
    np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), rmask)
    np.testing.assert_allclose(loss.item(), rloss, rtol=TOL)
    np.testing.assert_allclose(f.grad.cpu().numpy(), rgrad, rtol=1e-4, atol=1e-4 * np.abs(rgrad).max())
    if "nn" in sample:
        assert (nb[:, 1:1 + int(sample.split("nn")[1][0])] == n).any()                     # a shadow 'nn' column is part of the case


def test_tf_contrast_margin_temperature_and_internal_draws():
    """margin 'T<float>' sets the temperature (head.py:740-743); without rand_idx the draws come from the caller's generator, per cloud (:574-589)"""
    from contrastboundary_amd import heads
    n, k, d = 2400, 14, 16
    rng, feat, lab, nb, lens = _tf_radius_case(3, n, k, d)
    f = dev(feat)
    a = heads.tf_contrast(f, dev(lab), dev(nb), temperature=1.0, margin="ST.5", contrast="softnn")
    b = heads.tf_contrast(f, dev(lab), dev(nb), temperature=0.5, margin="S", contrast="softnn")
    assert a.item() == b.item()
    r = C.tf_contrast(feat, lab, nb, temperature=0.5, weight=0.1, separate=True, grad=False)[0]
    np.testing.assert_allclose(a.item(), r, rtol=TOL)
    g = torch.Generator(device="cuda").manual_seed(5)
    samples, roles, valid = heads.tf_sample_columns(dev(nb), "nn3-rand16R", batches_len=dev(lens), generator=g)
    assert samples.shape == (n, 1 + 3 + 16) and roles.tolist() == [1] * 3 + [3] * 16 and valid.shape == (n, 19)
    draws = samples[:, 4:].cpu().numpy()
    assert (draws[:lens[0]] < lens[0]).all() and (draws[lens[0]:] >= lens[0]).all() and (draws < n).all()   # every point draws from its own cloud
    want = (draws[:, :, None] != nb[:, None, 1:]).all(-1)
    np.testing.assert_array_equal(valid[:, 3:].cpu().numpy().astype(bool), want)
    g = torch.Generator(device="cuda").manual_seed(5)
    loss = heads.tf_contrast(f, dev(lab), dev(nb), temperature=0.8, sample="nn3-rand16R", batches_len=dev(lens), generator=g)
    rl = C.tf_contrast(feat, lab, nb, temperature=0.8, weight=0.1, sample="nn3-rand16R", rand_idx=[draws], grad=False)[0]
    np.testing.assert_allclose(loss.item(), rl, rtol=TOL)


def test_coincident_points_keep_the_reference_column_zero():
    """A point with a coincident twin: both are at distance 0 from the query and the reference's heap decides which one is column 0 — the
    column point_contrast drops as "the query itself" (heads.py:195-196).  The set-policy search the head uses leaves the order among
    equal distances free EXCEPT for that column, so the head's numbers stay the reference's on clouds with duplicated coordinates."""
    from contrastboundary_amd import heads, pointops
    rng = np.random.default_rng(11)
    n, K, d = 4096, 24, 32
    xyz = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    xyz[1::5] = xyz[0:n - 1:5][: len(xyz[1::5])]                      # every fifth point gets a coincident twin
    lab = (np.floor(xyz[:, 0] * 4) + 4 * np.floor(xyz[:, 1] * 3)).astype(np.int64) % 13
    feat = (rng.normal(size=(n, d)) * 0.5).astype(np.float32)
    off = np.int32([n])
    ridx, _ = O.knnquery(K, xyz, xyz, off, off)                       # the reference's order
    assert (ridx[:, 0] != np.arange(n)).any(), "the fixture must contain queries whose column 0 is their twin"
    xyz_d, off_d = dev(xyz), dev(off)
    sidx, _ = pointops.knnquery_raw(K, xyz_d, xyz_d, off_d, off_d, algo="set")
    np.testing.assert_array_equal(sidx.cpu().numpy()[:, 0], ridx[:, 0])
    np.testing.assert_array_equal(np.sort(sidx.cpu().numpy(), 1), np.sort(ridx, 1))
    f = dev(feat).requires_grad_(True)
    loss = heads.point_contrast(f, dev(lab), sidx, temperature=1.0, weight=0.1)
    loss.backward()
    rloss, rgrad, _ = C.point_contrast(feat, np.eye(13, dtype=np.float32)[lab], ridx, temperature=1.0, weight=0.1)
    np.testing.assert_allclose(loss.item(), rloss, rtol=TOL)
    np.testing.assert_allclose(f.grad.cpu().numpy(), rgrad, rtol=1e-4, atol=1e-4 * np.abs(rgrad).max())


@pytest.mark.parametrize("d", [8, 32])
def test_point_contrast_with_hub_targets_vs_oracle(d):
    """targets listed by far more than 64 pairs (several 64-entry chunks of the transposed table per target, most of them with a
    coefficient), beside targets nobody lists: the gather-form gradient against the oracle, and run-to-run identical"""
    from contrastboundary_amd import heads
    rng = np.random.default_rng(d)
    n, nsample = 6000, 24
    lab = rng.integers(0, 5, n).astype(np.int64)
    feat = (rng.normal(size=(n, d)) * 0.5).astype(np.float32)
    idx = rng.integers(100, n, size=(n, nsample)).astype(np.int32)        # rows 0..99 are listed by nobody ...
    idx[:, 0] = np.arange(n)                                              # (column 0 = the point itself, dropped by the head)
    idx[:, 1] = 7                                                         # ... except hub 7: listed by every point,
    idx[::3, 2] = 11                                                      # hub 11 by every third,
    idx[::50, 3] = 13                                                     # hub 13 by 120 points
    grads = []
    for _ in range(2):
        f = dev(feat).requires_grad_(True)
        loss, mask = heads.point_contrast(f, dev(lab), dev(idx), temperature=0.9, weight=0.1, return_mask=True)
        loss.backward()
        grads.append(f.grad.cpu().numpy())
    rloss, rgrad, rmask = C.point_contrast(feat, np.eye(5, dtype=np.float32)[lab], idx, temperature=0.9, weight=0.1)
    np.testing.assert_array_equal(mask.cpu().numpy().astype(bool), rmask)
    np.testing.assert_allclose(loss.item(), rloss, rtol=TOL)
    np.testing.assert_allclose(grads[0], rgrad, rtol=1e-4, atol=1e-4 * np.abs(rgrad).max())
    assert np.array_equal(grads[0], grads[1])


def test_point_contrast_beyond_the_transposed_table_limit_scatters_with_atomics():
    """more than 1 M rows: cbl_neighbor_transpose is unsupported there, the gradient's neighbour half falls back to the atomic scatter
    (cbl_contrast_pairs_backward_atomic).  (a) the fallback entry equals the gather entry at a size where both exist; (b) a 1.1 M-point
    head trains: loss and gradient against the oracle on a sample of rows."""
    import ctypes
    from contrastboundary_amd import _lib, heads, pointops
    L = _lib.lib()
    rng = np.random.default_rng(5)
    # (a) both entries on the same (coef, own)
    m, ns, d = 20000, 12, 16
    feat = rng.normal(size=(m, d)).astype(np.float32)
    nb = rng.integers(0, m, (m, ns)).astype(np.int32); nb[:, 0] = np.arange(m)
    coef = (rng.normal(size=(m, ns)) * (rng.uniform(size=(m, ns)) < 0.4)).astype(np.float32); coef[:, 0] = 0
    own = rng.normal(size=(m, d)).astype(np.float32)
    stats = np.float32([3.0, 500.0]); gl = np.float32([0.7])
    f_d, nb_d, c_d, o_d, s_d, g_d = dev(feat), dev(nb), dev(coef), dev(own), dev(stats), dev(gl)
    order, inv_start, inv_src = pointops.neighbor_transpose(nb_d, m)
    ga = torch.empty_like(f_d); gb = torch.empty_like(f_d)
    st = _lib.stream_of(f_d)
    _lib.check(L.cbl_contrast_pairs_backward(m, ns, d, _lib.ptr(f_d), _lib.ptr(c_d), _lib.ptr(o_d), _lib.ptr(order), _lib.ptr(inv_start), _lib.ptr(inv_src),
                                             _lib.ptr(s_d), _lib.ptr(g_d), ctypes.c_float(0.1), _lib.ptr(ga), st), "gather")
    _lib.check(L.cbl_contrast_pairs_backward_atomic(m, m, ns, d, _lib.ptr(f_d), _lib.ptr(c_d), _lib.ptr(o_d), _lib.ptr(nb_d), _lib.ptr(s_d), _lib.ptr(g_d),
                                                    ctypes.c_float(0.1), _lib.ptr(gb), st), "atomic")
    a, b = ga.cpu().numpy(), gb.cpu().numpy()
    np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-5 * np.abs(a).max())
    # (b) 1.1 M points through autograd
    m, ns, d = 1_100_000, 6, 4
    feat = rng.normal(size=(m, d)).astype(np.float32)
    lab = rng.integers(0, 3, m).astype(np.int64)
    nb = (np.arange(m)[:, None] + np.concatenate([[0], rng.integers(1, 50, ns - 1)])[None, :]) % m
    nb = nb.astype(np.int32)
    nb_d = dev(nb)
    assert pointops.neighbor_transpose(nb_d, m) is None                # no table at this size
    f = dev(feat).requires_grad_(True)
    loss = heads.point_contrast(f, dev(lab), nb_d, 1.0, 0.1)
    loss.backward()
    rloss, rgrad, _ = C.point_contrast(feat, np.eye(3, dtype=np.float32)[lab], nb, temperature=1.0, weight=0.1)
    assert abs(loss.item() - rloss) < 1e-4 * max(1.0, abs(rloss))
    np.testing.assert_allclose(f.grad.cpu().numpy(), rgrad, rtol=1e-4, atol=1e-4 * np.abs(rgrad).max())


def test_subscene_features_of_float_rows_match_reference():
    """get_subscene_features (basic_operators.py:16-50) on arbitrary per-point features, against the reference's own function run on CPU
    (tests/golden/gen_subscene_features_goldens.py): a mean over kr gathered rows — 1e-6 (the reference sums then divides, the kernel weights by 1/kr)"""
    from contrastboundary_amd.basic_operators import get_subscene_features
    F = np.load(os.path.join(os.path.dirname(__file__), "golden", "subscene_features.npz"))
    sl = make_stage_list("default")
    x = dev(F["x"])
    for i in range(5):
        got = get_subscene_features("up", i, sl, x, [4, 4, 4, 4])
        np.testing.assert_allclose(got.cpu().numpy(), F[f"stage{i}"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(get_subscene_features("up", 0, sl, x, [4, 4, 4, 4], extend=True).cpu().numpy(), F["stage0_extend"], rtol=1e-5, atol=1e-6)
    got, nidx, kr = get_subscene_features("up", 2, sl, x, [4, 4, 4, 4], kr=5, return_neighbor=True)
    np.testing.assert_allclose(got.cpu().numpy(), F["stage2_kr5"], rtol=1e-5, atol=1e-6)
    assert kr == 5 and nidx.numel() == got.shape[0] * 5
    xg = x.clone().requires_grad_(True)                              # differentiable w.r.t. the features, as the torch composite is
    get_subscene_features("up", 1, sl, xg, [4, 4, 4, 4]).sum().backward()
    assert abs(float(xg.grad.sum()) - sl["up"][1]["p_out"].shape[0] * x.shape[1]) < 1e-2
