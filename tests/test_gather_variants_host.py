"""CPU: the remaining forms of the neighbour gather / scatter operators, run from the host build of the whole library (tests/host_emul/full_library.py, wave
semantics) through their C entry points:
  - the `_ordered` forward entries (queryandgroup — the north-star kernel —, grouping, subtraction, aggregation, KPConv): a processing order changes the schedule,
    never the values — bit-identical to the plain entries (which tests/test_pointops_gather_host.py and test_local_aggregation_host.py hold to the oracle);
  - the scatter-adds of /root/reference/pytorch/lib/pointops/src/{grouping,interpolation,subtraction,aggregation}/*_cuda_kernel.cu as gathers over the transposed
    neighbour table: sums in ascending pair order = the reference loops run sequentially, stated with np.add.at in float32, bit for bit;
  - PosPool's feature gradient as a gather against its scatter form; interpolation weights (functions/pointops.py:171-174); ind_max_pool / ind_closest_pool
    (/root/reference/tensorflow/models/basic_operators.py:155-192); voxelize + crop order (pytorch/util/voxelize.py:38-56, data_util.py:62-64) against the oracle
    pinned by the reference's own module; the MFMA-vs-fmaf chain self-test of the fused layer."""
import ctypes

import numpy as np
import pytest

from oracle import voxelize_oracle as V
from tests import oracle_lib as O
from tests.host_emul import full_library

F = ctypes.c_float


@pytest.fixture(scope="module")
def host():
    L = full_library.load()
    for name in ("cbl_neighbor_transpose_workspace_bytes", "cbl_pospool_backward_csr_workspace_bytes", "cbl_voxelize_workspace_bytes"):
        getattr(L, name).restype = ctypes.c_size_t
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def scene(n, m, K, seed):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    q = np.ascontiguousarray(xyz[rng.choice(n, m, replace=False)]) if m < n else xyz
    idx, _ = O.knnquery(K, xyz, q, np.int32([n]), np.int32([m]))
    return xyz, q, np.ascontiguousarray(idx, np.int32), rng


def transposed(host, idx, n, order_dst=None):
    m, K = idx.shape
    inv_start, inv_src = np.full(n + 1, -1, np.int32), np.full(m * K, -1, np.int32)
    nbytes = host.cbl_neighbor_transpose_workspace_bytes(m, n, K)
    ws = np.zeros(nbytes + 64, np.uint8)
    assert host.cbl_neighbor_transpose(m, n, K, P(idx), None, P(order_dst), P(inv_start), P(inv_src), P(ws), ctypes.c_size_t(nbytes), None) == 0
    return inv_start, inv_src


@pytest.mark.parametrize("n,m,K,c", [(900, 900, 16, 32), (700, 300, 8, 64), (500, 500, 16, 6), (400, 400, 5, 20)])
def test_a_processing_order_never_changes_the_values(host, n, m, K, c):
    xyz, q, idx, rng = scene(n, m, K, seed=c)
    feat = rng.normal(size=(n, c)).astype(np.float32)
    order = rng.permutation(m).astype(np.int32)
    for use_xyz in (1, 0):
        w = c + 3 * use_xyz
        a, b = np.full((m, K, w), np.nan, np.float32), np.full((m, K, w), np.nan, np.float32)
        assert host.cbl_queryandgroup(m, K, c, use_xyz, P(xyz), P(q), P(feat), P(idx), P(a), None) == 0
        assert host.cbl_queryandgroup_ordered(m, K, c, use_xyz, P(xyz), P(q), P(feat), P(idx), P(order), P(b), None) == 0
        np.testing.assert_array_equal(bits(a), bits(b))
        ref = np.concatenate([xyz[idx] - q[:, None, :], feat[idx]], -1) if use_xyz else feat[idx]
        np.testing.assert_array_equal(bits(b), bits(ref))
    a, b = np.full((m, K, c), np.nan, np.float32), np.full((m, K, c), np.nan, np.float32)
    assert host.cbl_grouping_forward(m, K, c, P(feat), P(idx), P(a), None) == 0
    assert host.cbl_grouping_forward_ordered(m, K, c, P(feat), P(idx), P(order), P(b), None) == 0
    np.testing.assert_array_equal(bits(a), bits(b)); np.testing.assert_array_equal(bits(b), bits(feat[idx]))
    if m == n:
        f1 = rng.normal(size=(n, c)).astype(np.float32)
        assert host.cbl_subtraction_forward(n, K, c, P(f1), P(feat), P(idx), P(a), None) == 0
        assert host.cbl_subtraction_forward_ordered(n, K, c, P(f1), P(feat), P(idx), P(order), P(b), None) == 0
        np.testing.assert_array_equal(bits(a), bits(b)); np.testing.assert_array_equal(bits(b), bits(f1[:, None, :] - feat[idx]))
        if c % 8 == 0:
            wc = c // 8
            pos, wt = rng.normal(size=(n, K, c)).astype(np.float32), rng.normal(size=(n, K, wc)).astype(np.float32)
            oa, ob = np.zeros((n, c), np.float32), np.zeros((n, c), np.float32)
            assert host.cbl_aggregation_forward(n, K, c, wc, P(feat), P(pos), P(wt), P(idx), P(oa), None) == 0
            assert host.cbl_aggregation_forward_ordered(n, K, c, wc, P(feat), P(pos), P(wt), P(idx), P(order), P(ob), None) == 0
            np.testing.assert_array_equal(bits(oa), bits(ob))
            ref = np.zeros((n, c), np.float32)
            for k in range(K):                                           # the reference's loop over the neighbours, in float32
                ref += (feat[idx[:, k]] + pos[:, k]) * np.tile(wt[:, k], (1, 8))
            np.testing.assert_array_equal(bits(ob), bits(ref))


@pytest.mark.parametrize("K,C,KP,influence,closest", [(16, 64, 15, 1, 0), (20, 32, 15, 0, 1)])
def test_kpconv_under_a_processing_order(host, K, C, KP, influence, closest):
    n = 500
    xyz, q, idx, rng = scene(n, n, K, seed=K)
    idx[rng.random(idx.shape) < 0.1] = n                                # shadow neighbours
    feat, kp, kw = rng.normal(size=(n, C)).astype(np.float32), (rng.normal(size=(KP, 3)) * 0.05).astype(np.float32), rng.normal(size=(KP, C)).astype(np.float32)
    order = rng.permutation(n).astype(np.int32)
    a, b = np.full((n, C), np.nan, np.float32), np.full((n, C), np.nan, np.float32)
    assert host.cbl_kpconv_forward(n, n, K, C, KP, P(xyz), P(xyz), P(idx), P(feat), P(kp), P(kw), F(0.06), influence, closest, P(a), None) == 0
    assert host.cbl_kpconv_forward_ordered(n, n, K, C, KP, P(xyz), P(xyz), P(idx), P(feat), P(kp), P(kw), F(0.06), influence, closest, P(order), P(b), None) == 0
    np.testing.assert_array_equal(bits(a), bits(b))
    assert np.isfinite(a).all() and np.abs(a).max() > 0


@pytest.mark.parametrize("n,m,K,c", [(600, 600, 16, 32), (500, 200, 8, 12), (300, 300, 3, 7)])
def test_scatter_adds_as_gathers_over_the_transposed_table(host, n, m, K, c):
    xyz, q, idx, rng = scene(n, m, K, seed=K + c)
    idx[m // 2] = 5                                                     # K pairs of one source on one target
    order_dst = rng.permutation(n).astype(np.int32)
    for od in (None, order_dst):
        inv_start, inv_src = transposed(host, idx, n, od)
        # K4 on a column slice of wider rows (queryandgroup's (m, K, 3 + c) gradient, pointops.py:90-98)
        go = rng.normal(size=(m, K, 3 + c)).astype(np.float32)
        gi = np.full((n, c), np.nan, np.float32)
        assert host.cbl_grouping_backward_csr_rows(n, c, 3 + c, 3, P(go), P(od), P(inv_start), P(inv_src), P(gi), None) == 0
        ref = np.zeros((n, c), np.float32)
        np.add.at(ref, idx.reshape(-1), go[:, :, 3:].reshape(-1, c))      # grouping_cuda_kernel.cu:16-25 run sequentially
        np.testing.assert_array_equal(bits(gi), bits(ref))
        # K10's grad_input (aggregation_cuda_kernel.cu:22-39) and, with three columns and one weight per pair, K6 (interpolation_cuda_kernel.cu:20-33)
        for wc in ([c // 8] if c % 8 == 0 else []) + [1]:
            rows, wt = rng.normal(size=(m, c)).astype(np.float32), rng.normal(size=(m, K, wc)).astype(np.float32)
            gi = np.full((n, c), np.nan, np.float32)
            assert host.cbl_weighted_scatter_csr(n, K, c, wc, P(rows), P(wt), P(od), P(inv_start), P(inv_src), P(gi), None) == 0
            ref = np.zeros((n, c), np.float32)
            np.add.at(ref, idx.reshape(-1), (rows[:, None, :] * np.tile(wt, (1, 1, c // wc))).reshape(-1, c))
            np.testing.assert_array_equal(bits(gi), bits(ref))
        # K8 (subtraction_cuda_kernel.cu:18-30): grad_input1 accumulated, grad_input2 written
        go = rng.normal(size=(m, K, c)).astype(np.float32)
        g1, g2 = rng.normal(size=(m, c)).astype(np.float32), np.full((n, c), np.nan, np.float32)
        r1 = g1.copy()
        assert host.cbl_subtraction_backward_csr(m, n, K, c, P(go), P(od), P(inv_start), P(inv_src), P(g1), P(g2), None) == 0
        for k in range(K):
            r1 += go[:, k]
        r2 = np.zeros((n, c), np.float32)
        np.add.at(r2, idx.reshape(-1), -go.reshape(-1, c))
        np.testing.assert_array_equal(g2, r2)                            # (as values: a target no pair lists holds -0.0 here, +0.0 there)
        np.testing.assert_allclose(g1, r1, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,Ka,Kb", [(900, 36, 16), (700, 8, 36), (300, 16, 16), (130, 3, 40)])
def test_two_tables_of_one_geometry_transposed_together(host, n, Ka, Kb):
    """cbl_neighbor_transpose_pair: the K = 8 / 16 table of a stage's blocks and the K = 36 table of its CBL head (blocks.py:34-35, heads.py:190-196) by the same four
    launches — byte for byte the outputs of two cbl_neighbor_transpose calls, with and without a processing order, shadow entries left out"""
    host.cbl_neighbor_transpose_pair_workspace_bytes.restype = ctypes.c_size_t
    xyz, _, idx_a, rng = scene(n, n, Ka, seed=n)
    idx_b, _ = O.knnquery(Kb, xyz, xyz, np.int32([n]), np.int32([n]))
    idx_b = np.ascontiguousarray(idx_b, np.int32)
    idx_a[rng.random(idx_a.shape) < 0.05] = n                           # shadow / padding entries
    idx_b[n // 3] = 7                                                   # Kb pairs of one source on one target
    order = rng.permutation(n).astype(np.int32)
    for od in (None, order):
        sa, ia = transposed(host, idx_a, n, od)
        sb, ib = transposed(host, idx_b, n, od)
        nbytes = host.cbl_neighbor_transpose_pair_workspace_bytes(n, n, Ka, Kb)
        assert nbytes >= host.cbl_neighbor_transpose_workspace_bytes(n, n, Ka) + host.cbl_neighbor_transpose_workspace_bytes(n, n, Kb)
        ws = np.zeros(nbytes + 64, np.uint8)
        pa, qa = np.full(n + 1, -1, np.int32), np.full(n * Ka, -1, np.int32)
        pb, qb = np.full(n + 1, -1, np.int32), np.full(n * Kb, -1, np.int32)
        assert host.cbl_neighbor_transpose_pair(n, n, Ka, P(idx_a), Kb, P(idx_b), P(od), P(od), P(pa), P(qa), P(pb), P(qb), P(ws), ctypes.c_size_t(nbytes), None) == 0
        np.testing.assert_array_equal(pa, sa); np.testing.assert_array_equal(pb, sb)
        np.testing.assert_array_equal(qa[:sa[n]], ia[:sa[n]]); np.testing.assert_array_equal(qb[:sb[n]], ib[:sb[n]])
        assert host.cbl_neighbor_transpose_pair(n, n, Ka, P(idx_a), Kb, P(idx_b), P(od), P(od), P(pa), P(qa), P(pb), P(qb), P(ws), ctypes.c_size_t(nbytes - 256), None) == -2


@pytest.mark.parametrize("K,C,embedding,reduction", [(16, 36, 1, 1), (20, 24, 0, 0)])
def test_pospool_feature_gradient_as_a_gather(host, K, C, embedding, reduction):
    n = 400
    xyz, q, idx, rng = scene(n, n, K, seed=C)
    idx[rng.random(idx.shape) < 0.15] = n
    pad = np.int32([int(idx.max())])
    go = rng.normal(size=(n, C)).astype(np.float32)
    feat = rng.normal(size=(n, C)).astype(np.float32)
    ga = np.zeros((n, C), np.float32)
    assert host.cbl_pospool_backward(n, n, K, C, P(xyz), P(xyz), P(idx), P(feat), F(0.2), embedding, reduction, P(pad), P(go), P(ga), None) == 0
    inv_start, inv_src = transposed(host, idx, n)
    gb = np.full((n, C), np.nan, np.float32)
    nbytes = host.cbl_pospool_backward_csr_workspace_bytes(n)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = host.cbl_pospool_backward_csr(n, n, K, C, P(xyz), P(xyz), P(idx), F(0.2), embedding, reduction, P(pad), P(go), None, P(inv_start), P(inv_src), P(gb),
                                       P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    assert np.abs(ga).max() > 0
    np.testing.assert_allclose(gb, ga, rtol=1e-5, atol=1e-5 * np.abs(ga).max())


def test_interpolation_weights_and_index_pools(host):
    rng = np.random.default_rng(0)
    d2 = rng.uniform(0, 2, (500, 3)).astype(np.float32); d2[::50, 0] = 0
    w, d = np.full((500, 3), np.nan, np.float32), np.full((500, 3), np.nan, np.float32)
    assert host.cbl_interpolation_weights(500, 3, P(d2), P(w), P(d), None) == 0
    dist = np.sqrt(d2); r = np.float32(1.0) / (dist + np.float32(1e-8))
    np.testing.assert_allclose(d, dist, rtol=1e-6)
    np.testing.assert_allclose(w, r / r.sum(1, keepdims=True), rtol=1e-5)
    assert host.cbl_interpolation_weights(500, 3, P(d2), P(w), None, None) == 0
    # ind_max_pool / ind_closest_pool (basic_operators.py:155-192): pad index n1 selects the shadow row
    n1, n2, k, dch = 300, 120, 9, 40
    x = rng.normal(size=(n1, dch)).astype(np.float32)
    inds = rng.integers(0, n1 + 1, (n2, k)).astype(np.int32)
    inds[3] = n1                                                        # a row of padding only
    scratch, out = np.zeros(dch, np.uint32), np.full((n2, dch), np.nan, np.float32)
    assert host.cbl_ind_max_pool(n1, n2, k, dch, P(x), P(inds), P(scratch), P(out), None) == 0
    xp = np.concatenate([x, x.min(0, keepdims=True)])
    np.testing.assert_array_equal(bits(out), bits(xp[inds].max(1)))
    assert host.cbl_ind_closest_pool(n1, n2, k, dch, P(x), P(inds), P(out), None) == 0
    xz = np.concatenate([x, np.zeros((1, dch), np.float32)])
    np.testing.assert_array_equal(bits(out), bits(xz[inds[:, 0]]))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_voxelize_and_crop_order(host, dtype):
    rng = np.random.default_rng(2)
    n = 3000
    coord = rng.uniform(0, 3, (n, 3)).astype(dtype)
    coord[::7] = coord[5]                                               # coincident points: one voxel, the stable order decides
    vs = 0.2
    key, idx_sort, start, count = V.voxelize(coord, vs)
    ks, isort, st, cn, nv = np.zeros(n, np.uint64), np.full(n, -1, np.int32), np.full(n, -1, np.int32), np.full(n, -1, np.int32), np.full(1, -1, np.int32)
    nbytes = host.cbl_voxelize_workspace_bytes(n)
    ws = np.zeros(nbytes + 64, np.uint8)
    assert host.cbl_voxelize(n, 1 if dtype == np.float64 else 0, P(coord), ctypes.c_double(vs), P(ks), P(isort), P(st), P(cn), P(nv), P(ws), ctypes.c_size_t(nbytes), None) == 0
    v = int(nv[0])
    assert v == len(count)
    np.testing.assert_array_equal(ks, key[idx_sort]); np.testing.assert_array_equal(isort, idx_sort)
    np.testing.assert_array_equal(st[:v], start); np.testing.assert_array_equal(cn[:v], count)
    order = np.full(n, -1, np.int32)
    assert host.cbl_crop_order(n, 1 if dtype == np.float64 else 0, P(coord), 123, P(order), P(ws), ctypes.c_size_t(nbytes), None) == 0
    np.testing.assert_array_equal(order, V.crop_order(coord, 123))


def test_mfma_tile_equals_the_ordered_fmaf_chain(host):
    """the numerical assumption behind the fused layer's agreeing ReLU masks (pt_layer.hip): one 16x16x4 f32 MFMA tile next to the k-ordered fmaf chain.
    Here both run through the emulated builtin, so this holds the entry point and the tile's lane layout, not the hardware's rounding
    (tests/test_gpu_pt_layer.py holds that on the device)."""
    rng = np.random.default_rng(1)
    A, B, C = rng.normal(size=(16, 4)).astype(np.float32), rng.normal(size=(4, 16)).astype(np.float32), rng.normal(size=(16, 16)).astype(np.float32)
    d1, d2 = np.full((16, 16), np.nan, np.float32), np.full((16, 16), np.nan, np.float32)
    assert host.cbl_pt_layer_selftest_chain(P(A), P(B), P(C), P(d1), P(d2), None) == 0
    np.testing.assert_array_equal(bits(d1), bits(d2))
    np.testing.assert_allclose(d1, A.astype(np.float64) @ B.astype(np.float64) + C, rtol=1e-5, atol=1e-5)
