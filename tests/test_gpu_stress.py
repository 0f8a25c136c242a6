"""Large / awkward shapes: nothing here is compared with the O(n*m) oracle — the checks are size-independent properties
(sortedness, self-neighbour, distance recomputation, uniqueness, agreement between independent kernels)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def P():
    from contrastboundary_amd import pointops
    return pointops


def test_knn_one_million_points_many_clouds(P):
    rng = np.random.default_rng(0)
    n = 1_000_000
    xyz = rng.uniform(0, 20, (n, 3)).astype(np.float32)
    cuts = np.sort(rng.choice(np.arange(1, n), 63, replace=False))
    cuts[:8] = np.arange(1, 9) * 3                                      # a few 3-point clouds (fewer points than K)
    offset = np.concatenate([np.sort(cuts), [n]]).astype(np.int32)
    x, o = dev(xyz), dev(offset)
    for K, algo in ((16, "auto"), (36, "set"), (8, "anytie")):
        idx, d2 = P.knnquery_raw(K, x, x, o, o, algo=algo)
        d2c, ic = d2.cpu().numpy(), idx.cpu().numpy().astype(np.int64)
        big = np.searchsorted(offset, np.arange(n), side="right")       # cloud of each point
        lens = np.diff(np.concatenate([[0], offset]))
        full = lens[big] >= K
        assert np.all(np.diff(d2c[full], axis=1) >= 0)                  # ascending
        assert np.all(d2c[full][:, 0] == 0)                             # the query itself is its nearest support
        sel = rng.choice(np.nonzero(full)[0], 20000, replace=False)
        dd = (xyz[ic[sel]] - xyz[sel][:, None, :]) ** 2
        np.testing.assert_array_equal(((dd[..., 0] + dd[..., 1]) + dd[..., 2]).astype(np.float32), d2c[sel])
        assert np.all(big[ic[sel]] == big[sel][:, None])                # neighbours come from the query's own cloud


def test_knn_k_extremes(P):
    rng = np.random.default_rng(1)
    xyz = dev(rng.uniform(0, 1, (6000, 3)).astype(np.float32)); o = dev(np.int32([6000]))
    ref, rd = P.knnquery_raw(1024, xyz, xyz, o, o)
    for K in (1, 2, 63, 64, 65, 127, 1024):
        idx, d2 = P.knnquery_raw(K, xyz, xyz, o, o)
        assert torch.equal(d2, rd[:, :K].contiguous())                  # a prefix of the K = 1024 answer (tie-free data)
        assert torch.equal(idx, ref[:, :K].contiguous())
    with pytest.raises(ValueError):
        P.knnquery_raw(1025, xyz, xyz, o, o)


def test_fps_large_and_beyond_the_bucket_kernel(P):
    rng = np.random.default_rng(2)
    for n, m in ((200_000, 2000), (100_000, 25_000)):                   # > 131072: streaming kernel; <=: bucket kernel
        xyz = rng.normal(0, 1, (n, 3)).astype(np.float32)
        idx = P.furthestsampling(dev(xyz), dev(np.int32([n])), dev(np.int32([m]))).cpu().numpy()
        assert idx[0] == 0 and len(np.unique(idx)) == m
        # greedy property on a prefix: sample j+1 is the support furthest from samples 0..j
        s = xyz[idx[:64]]
        t = np.full(n, np.float32(1e10))
        for j in range(63):
            d = xyz - s[j]; d = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)
            t = np.minimum(t, d)
            assert t[idx[j + 1]] == t.max()


def test_radius_and_subsampling_at_a_million_points():
    from contrastboundary_amd import tf_ops, synthetic as S
    xyz, _ = S.s_room(1_000_000, seed=1, scale=8.0)
    x = dev(xyz); lens = dev(np.int32([len(xyz)]))
    sub, sub_l = tf_ops.tf_batch_subsampling(x, lens, 0.1)
    assert int(sub_l.sum()) == sub.shape[0] and 1000 < sub.shape[0] < len(xyz)
    nb = tf_ops.tf_batch_neighbors(sub, x, sub_l, lens, 0.1, 40)
    nbc = nb.cpu().numpy().astype(np.int64)
    q = sub.cpu().numpy()
    real = nbc < len(xyz)
    rows = np.nonzero(real.any(1))[0][:5000]
    for r in rows[::50]:
        js = nbc[r][real[r]]
        d = np.linalg.norm(xyz[js] - q[r], axis=1)
        assert np.all(d < 0.1 + 1e-6) and np.all(np.diff(d) >= -1e-6)


@pytest.mark.parametrize("order", ["shuffled", "sorted"])
def test_reference_order_replay_in_large_clouds(order):
    """clouds above 65536 supports: the replay of tied queries takes the supports beyond the first 32768 from the search grid (a tight
    bound around the query) instead of scanning the cloud; the reference's heap order must come out all the same"""
    import torch
    from contrastboundary_amd import pointops
    from tests import oracle_lib as O
    rng = np.random.default_rng(11)
    n_big, n_small, K = 150000, 9000, 16
    base = rng.uniform(0, 8, (n_big - 400, 3)).astype(np.float32)
    big = np.concatenate([base, base[rng.choice(len(base), 400, replace=False)]])      # 400 coincident pairs: zero distances, ties inside the lists
    if order == "sorted":
        big = big[np.lexsort((big[:, 0], big[:, 1], big[:, 2]))]                        # spatially sorted: long heap histories
    else:
        big = big[rng.permutation(n_big)]
    small = rng.uniform(0, 2, (n_small, 3)).astype(np.float32)
    xyz = np.concatenate([small, big]); off = np.int32([n_small, n_small + n_big])
    p = torch.from_numpy(xyz).cuda(); o = torch.from_numpy(off).cuda()
    idx, d2 = pointops.knnquery_raw(K, p, p, o, o)                                     # reference order
    torch.cuda.synchronize()
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    tied = np.nonzero((d2[:, 1:] == d2[:, :-1]).any(1))[0]
    assert (tied >= n_small).sum() >= 300                                              # the tied queries are in the big cloud
    pick = np.concatenate([tied[tied >= n_small][:60], rng.integers(n_small, n_small + n_big, 20), rng.integers(0, n_small, 10)])
    q = xyz[pick]
    # the oracle needs per-cloud query blocks: queries of cloud 0 first
    order_q = np.argsort(pick >= n_small, kind="stable"); pick, q = pick[order_q], q[order_q]
    qoff = np.int32([(pick < n_small).sum(), len(pick)])
    ri, rd = O.knnquery(K, xyz, q, off, qoff)
    np.testing.assert_array_equal(idx[pick], ri)
    np.testing.assert_array_equal(d2[pick], rd)
