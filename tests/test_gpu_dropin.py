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


def test_scalars_as_zero_dim_tensors(pc):
    """heads.py:186/192 passes nsample = self.nsample[i] (a 0-dim tensor), basic_operators.py:22 kr = torch.prod(...): pybind takes them
    through __index__, so must the drop-in"""
    rng = np.random.default_rng(1)
    n, K = 2500, 16
    xyz_h = rng.uniform(size=(n, 3)).astype(np.float32)
    off_h = np.int32([n])
    xyz, offset = torch.from_numpy(xyz_h).cuda(), torch.from_numpy(off_h).cuda()
    nsample = torch.tensor([8, K])[1]                                 # 0-dim int64 tensor, as config.nsample yields
    idx = torch.cuda.IntTensor(n, K).zero_(); dist2 = torch.cuda.FloatTensor(n, K).zero_()
    pc.knnquery_cuda(n, nsample, xyz, xyz, offset, offset, idx, dist2)
    ridx, _ = O.knnquery(K, xyz_h, xyz_h, off_h, off_h)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    kr = torch.prod(torch.tensor([4, 4]))                             # basic_operators.py:22
    idx2 = torch.cuda.IntTensor(n, 16).zero_(); d2 = torch.cuda.FloatTensor(n, 16).zero_()
    pc.knnquery_cuda(torch.tensor(n), kr, xyz, xyz, offset, offset, idx2, d2)
    np.testing.assert_array_equal(idx2.cpu().numpy(), ridx)


def test_boundary_rejects_what_the_kernels_cannot_read(pc):
    """SURVEY 8(b) "Error convention": wrong dtype / device / layout / shape raise before any pointer is taken"""
    n, K = 2048, 8
    xyz = torch.rand(n, 3, device="cuda")
    offset = torch.tensor([n], dtype=torch.int32, device="cuda")
    idx = torch.zeros(n, K, dtype=torch.int32, device="cuda"); dist2 = torch.zeros(n, K, device="cuda")
    pc.knnquery_cuda(n, K, xyz, xyz, offset, offset, idx, dist2)      # the well-formed call
    with pytest.raises(TypeError, match="offset must be torch.int32"):
        pc.knnquery_cuda(n, K, xyz, xyz, offset.long(), offset, idx, dist2)
    with pytest.raises(ValueError, match="xyz must be contiguous"):
        pc.knnquery_cuda(n, K, torch.rand(3, n, device="cuda").t(), xyz, offset, offset, idx, dist2)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        pc.knnquery_cuda(n, K, xyz.cpu(), xyz, offset, offset, idx, dist2)
    with pytest.raises(TypeError, match="xyz must be torch.float32"):
        pc.knnquery_cuda(n, K, xyz.double(), xyz, offset, offset, idx, dist2)
    with pytest.raises(ValueError, match="idx has shape"):
        pc.knnquery_cuda(n, K, xyz, xyz, offset, offset, torch.zeros(n, K + 1, dtype=torch.int32, device="cuda"), dist2)
    with pytest.raises(TypeError, match="idx must be torch.int32"):
        pc.grouping_forward_cuda(n, K, 4, torch.rand(n, 4, device="cuda"), idx.long(), torch.zeros(n, K, 4, device="cuda"))
    with pytest.raises(ValueError, match="output has shape"):
        pc.grouping_forward_cuda(n, K, 4, torch.rand(n, 4, device="cuda"), idx, torch.zeros(n, K, 5, device="cuda"))
    torch.cuda.synchronize()


def test_one_scratch_per_stream(pc):
    n, K = 4096, 16
    xyz = torch.rand(n, 3, device="cuda")
    offset = torch.tensor([n], dtype=torch.int32, device="cuda")
    out = []
    for st in (torch.cuda.current_stream(), torch.cuda.Stream()):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            idx = torch.zeros(n, K, dtype=torch.int32, device="cuda"); dist2 = torch.zeros(n, K, device="cuda")
            pc.knnquery_cuda(n, K, xyz, xyz, offset, offset, idx, dist2)
            out.append(idx)
    torch.cuda.synchronize()
    assert torch.equal(out[0], out[1])
    assert len({key[1] for key in pc._ws}) >= 2                       # keyed by (device, stream)
