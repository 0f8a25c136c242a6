"""GPU: the drop-in `pointops_cuda` module, driven exactly the way the reference's pointops.py drives the CUDA extension
(legacy typed constructors, caller-allocated zeroed outputs: pointops.py:21-23, 40-42, 57-58, 72-73, 195-196)."""
import os
import sys

import numpy as np
import pytest
import torch

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pc():
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "contrastboundary_amd", "dropin")
    sys.path.insert(0, d)
    import pointops_cuda
    yield pointops_cuda
    sys.path.remove(d)


def test_reference_call_patterns(pc):
    rng = np.random.default_rng(0)
    n, m, K, c = 3000, 750, 16, 32
    xyz_h = rng.uniform(size=(n, 3)).astype(np.float32)
    off_h, noff_h = np.int32([1000, 3000]), np.int32([250, 750])
    q_h = np.concatenate([xyz_h[0:1000:4], xyz_h[1000:3000:4]])
    xyz, q = torch.from_numpy(xyz_h).cuda(), torch.from_numpy(q_h).cuda()
    offset, new_offset = torch.from_numpy(off_h).cuda(), torch.from_numpy(noff_h).cuda()
    # KNNQuery.forward, pointops.py:38-43
    idx = torch.cuda.IntTensor(m, K).zero_()
    dist2 = torch.cuda.FloatTensor(m, K).zero_()
    pc.knnquery_cuda(m, K, xyz, q, offset, new_offset, idx, dist2)
    ridx, rd2 = O.knnquery(K, xyz_h, q_h, off_h, noff_h)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(dist2.cpu().numpy().view(np.uint32), rd2.view(np.uint32))
    # FurthestSampling.forward, :17-24: n_max arrives as a 0-dim tensor / python int mix
    n_max = offset[0]
    for i in range(1, 2):
        n_max = max(offset[i] - offset[i - 1], n_max)
    fidx = torch.cuda.IntTensor(new_offset[1].item()).zero_()
    tmp = torch.cuda.FloatTensor(n).fill_(1e10)
    pc.furthestsampling_cuda(2, n_max, xyz, offset, new_offset, tmp, fidx)
    rf, _ = O.furthestsampling(xyz_h, off_h, noff_h)
    np.testing.assert_array_equal(fidx.cpu().numpy(), rf)
    # Grouping, :55-74
    feat_h = rng.normal(size=(n, c)).astype(np.float32); feat = torch.from_numpy(feat_h).cuda()
    out = torch.cuda.FloatTensor(m, K, c)
    pc.grouping_forward_cuda(m, K, c, feat, idx, out)
    np.testing.assert_array_equal(out.cpu().numpy(), feat_h[ridx])
    go = torch.ones(m, K, c, device="cuda"); gi = torch.cuda.FloatTensor(n, c).zero_()
    pc.grouping_backward_cuda(m, K, c, go, idx, gi)
    np.testing.assert_allclose(gi.cpu().numpy()[:, 0], np.bincount(ridx.reshape(-1), minlength=n), atol=1e-4)
    # Interpolation, :188-211
    w_h = rng.uniform(size=(n, 3)).astype(np.float32); iidx_h = rng.integers(0, m, (n, 3)).astype(np.int32)
    src_h = rng.normal(size=(m, c)).astype(np.float32)
    o2 = torch.cuda.FloatTensor(n, c).zero_()
    pc.interpolation_forward_cuda(n, c, 3, torch.from_numpy(src_h).cuda(), torch.from_numpy(iidx_h).cuda(), torch.from_numpy(w_h).cuda(), o2)
    np.testing.assert_array_equal(o2.cpu().numpy(), O.interpolation_forward(src_h, iidx_h, w_h))
    # Subtraction / Aggregation, :110-157
    sidx_h = rng.integers(0, n, (n, K)).astype(np.int32); b_h = rng.normal(size=(n, c)).astype(np.float32)
    so = torch.cuda.FloatTensor(n, K, c).zero_()
    pc.subtraction_forward_cuda(n, K, c, feat, torch.from_numpy(b_h).cuda(), torch.from_numpy(sidx_h).cuda(), so)
    np.testing.assert_array_equal(so.cpu().numpy(), O.subtraction_forward(feat_h, b_h, sidx_h))
    pos_h = rng.normal(size=(n, K, c)).astype(np.float32); ww_h = rng.normal(size=(n, K, 4)).astype(np.float32)
    ao = torch.cuda.FloatTensor(n, c).zero_()
    pc.aggregation_forward_cuda(n, K, c, 4, feat, torch.from_numpy(pos_h).cuda(), torch.from_numpy(ww_h).cuda(), torch.from_numpy(sidx_h).cuda(), ao)
    np.testing.assert_array_equal(ao.cpu().numpy(), O.aggregation_forward(feat_h, pos_h, ww_h, sidx_h))
    torch.cuda.synchronize()
