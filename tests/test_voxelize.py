"""a13 voxelize: CPU oracle vs goldens from the reference's own voxelize.py; GPU kernels vs the oracle."""
import os

import numpy as np
import pytest

from oracle import voxelize_oracle as V

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "voxelize.npz"))
CASES = sorted({k.split("/")[0] for k in G.files})


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference(case):
    coord, vs = G[f"{case}/coord"], float(G[f"{case}/voxel"])
    key, idx_sort, start, count = V.voxelize(coord, vs)
    np.testing.assert_array_equal(key, G[f"{case}/key"])
    np.testing.assert_array_equal(count, G[f"{case}/count"])
    ref_sort = G[f"{case}/idx_sort"]
    np.testing.assert_array_equal(key[idx_sort], key[ref_sort])                       # same sorted key sequence
    for s, c in zip(start[:500], count[:500]):                                         # same index SET per voxel (quicksort order is free)
        assert set(idx_sort[s:s + c]) == set(ref_sort[s:s + c])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_voxelize_and_crop(case):
    import torch
    from contrastboundary_amd import voxelize as VZ
    coord, vs = G[f"{case}/coord"], float(G[f"{case}/voxel"])
    key, idx_sort, start, count = V.voxelize(coord, vs)
    c = torch.from_numpy(coord).cuda()
    gi, gc = VZ.voxelize(c, vs, mode=1)
    np.testing.assert_array_equal(gi.cpu().numpy(), idx_sort)
    np.testing.assert_array_equal(gc.cpu().numpy(), count)
    rand = np.random.default_rng(0).integers(0, count.max(), count.size)
    gu = VZ.voxelize(c, vs, mode=0, rand=torch.from_numpy(rand))
    np.testing.assert_array_equal(gu.cpu().numpy(), idx_sort[start + rand % count])    # voxelize.py:49-51
    order = VZ.crop_nearest(c, 123, 2000).cpu().numpy()
    np.testing.assert_array_equal(order, V.crop_order(coord, 123)[:2000])


def test_oracle_test_time_crops_cover_the_cloud():
    rng = np.random.default_rng(3)
    coord = rng.uniform(0, 4, (3000, 3)).astype(np.float32)
    crops = V.test_time_crops(coord, 800, rng.random(3000) * 1e-3)
    assert len(crops) >= 4 and all(len(c) == 800 for c in crops)
    assert np.array_equal(np.unique(np.concatenate(crops)), np.arange(3000))          # every point covered (the loop's exit condition)
    first = crops[0]
    d0 = ((coord[first] - coord[first[0]]) ** 2).sum(1)
    assert (np.diff(d0) >= 0).all()                                                    # nearest first, centre at position 0


@pytest.mark.gpu
def test_gpu_test_time_crops_equal_the_oracle():
    import torch
    from contrastboundary_amd import voxelize as VZ
    rng = np.random.default_rng(5)
    coord = rng.uniform(0, 5, (6000, 3)).astype(np.float32)
    pot = rng.random(6000) * 1e-3
    want = V.test_time_crops(coord, 1500, pot)
    got = VZ.test_time_crops(torch.from_numpy(coord).cuda(), 1500, torch.from_numpy(pot))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g.cpu().numpy(), w)


def test_synthetic_rooms_fail_fast_and_scale():
    """the scene generator refuses a point count its room cannot hold (instead of doubling its sample 8 times), and the bench scenes
    above 100k points grow the room"""
    from contrastboundary_amd import hotpath, synthetic as S
    with pytest.raises(ValueError):
        S.s_room(400000, seed=0)
    a = hotpath.Scene.synthetic_numpy(150000, 4, seed=1)
    assert a["xyz"].shape == (150000, 3) and a["labels"].shape == (150000,) and a["xyz"].dtype == np.float32
    b = hotpath.Scene.synthetic_numpy(4096, 4, seed=1)
    assert b["xyz"].shape == (4096, 3) and float(b["xyz"].min()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("smooth", [None, 0.8])
def test_cumulate_probs_matches_the_reference_expression(smooth):
    """tool/test.py:330-352.  The reference's update is numpy / torch-CPU semantics of `a[inds] += b` with duplicated inds (overlapping crops
    of one batch): gather, add, indexed assignment — the LAST row of a duplicated point counts.  Bit-exact against that expression."""
    import torch
    from contrastboundary_amd import voxelize as VZ
    rng = np.random.default_rng(3)
    n, ncls = 5000, 13
    cum = VZ.init_cumulate_dict(n, ncls, probs_last=True)
    ref = np.zeros((n, ncls), np.float32); ref_last = np.zeros((n, ncls), np.float32)
    for batch in range(3):
        crops = [rng.choice(n, 1800, replace=False) for _ in range(3)]            # three overlapping crops per batch
        inds = np.concatenate(crops)
        pred = rng.normal(size=(inds.size, ncls)).astype(np.float32)
        VZ.cumulate_probs(cum, torch.from_numpy(pred).cuda(), inds, smooth=smooth)
        if smooth is None:
            ref[inds, ...] += pred                                                # the reference's line 333, on the CPU
        else:
            ref[inds, ...] = np.float32(smooth) * ref[inds, ...] + np.float32(1 - smooth) * pred     # :335
        ref_last[inds, ...] = pred                                                # :350
    if smooth is None:
        np.testing.assert_array_equal(cum["probs"].cpu().numpy(), ref)
    else:
        np.testing.assert_allclose(cum["probs"].cpu().numpy(), ref, rtol=1e-6, atol=1e-7)     # float(1 - smooth) on the device vs numpy's float32(1 - smooth)
    np.testing.assert_array_equal(cum["probs_last"].cpu().numpy(), ref_last)
