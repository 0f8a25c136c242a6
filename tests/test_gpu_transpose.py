"""Transposed neighbour table (cbl_neighbor_transpose) and the scatter-add backward passes as gathers over it.
The table is integer work: bit-exact against a numpy restatement (stable sort of the pairs by target).  The gather-form grouping backward
sums a target's pairs in ascending order, i.e. in the order of the reference loop run sequentially (grouping_cuda_kernel.cu:16-25):
bit-exact against the CPU oracle, not merely within 1e-4."""
import ctypes

import numpy as np
import pytest
import torch

from contrastboundary_amd import _lib, pointops
from contrastboundary_amd import synthetic as S
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _i(v):
    return ctypes.c_int(int(v))


def transpose_gpu(idx, n, order=None, order_src=None):
    m, ns = idx.shape
    L = _lib.lib()
    idx_d = dev(idx.astype(np.int32))
    o_d = None if order is None else dev(order.astype(np.int32))
    os_d = o_d if order_src is None else dev(order_src.astype(np.int32))
    inv_start = torch.full((n + 1,), -7, dtype=torch.int32, device="cuda")
    inv_src = torch.full((max(m * ns, 1),), -7, dtype=torch.int32, device="cuda")
    need = L.cbl_neighbor_transpose_workspace_bytes(_i(m), _i(n), _i(ns))
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device="cuda")
    _lib.check(L.cbl_neighbor_transpose(_i(m), _i(n), _i(ns), _lib.ptr(idx_d), _lib.ptr(os_d), _lib.ptr(o_d), _lib.ptr(inv_start), _lib.ptr(inv_src),
                                        _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(idx_d)), "cbl_neighbor_transpose")
    torch.cuda.synchronize()
    return inv_start.cpu().numpy(), inv_src.cpu().numpy()


def transpose_numpy(idx, n, order=None):
    flat = idx.reshape(-1).astype(np.int64)
    keep = np.nonzero((flat >= 0) & (flat < n))[0]
    rank = np.arange(n) if order is None else np.argsort(order)           # rank[order[r]] = r
    key = rank[flat[keep]]
    perm = np.argsort(key, kind="stable")                                 # by target slot, pairs ascending inside a slot
    counts = np.bincount(key, minlength=n)
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return start, keep[perm].astype(np.int32)


def check(idx, n, order=None, order_src=None):
    gs, gsrc = transpose_gpu(idx, n, order, order_src)
    rs, rsrc = transpose_numpy(idx, n, order)
    assert np.array_equal(gs, rs)
    assert np.array_equal(gsrc[:rs[-1]], rsrc)


@pytest.mark.parametrize("m,n,ns", [(1000, 1000, 16), (4096, 4096, 36), (70, 50, 8), (5000, 777, 3), (1, 1, 1), (64, 64, 65), (3000, 3000, 1)])
def test_transpose_random(m, n, ns):
    rng = np.random.default_rng(m + ns)
    check(rng.integers(0, n, size=(m, ns)), n)
    perm = rng.permutation(n)
    if m == n:
        check(rng.integers(0, n, size=(m, ns)), n, order=perm)


def test_transpose_of_a_large_table():
    """more than 2048 target tiles: the scan of the bin sizes runs as a launch of its own (below that every source tile scans them itself)"""
    rng = np.random.default_rng(11)
    m = n = 200000
    ns = 8
    base = np.arange(m)[:, None]
    idx = (base + rng.integers(-3000, 3000, size=(m, ns))) % n              # neighbours near the source in index space, a few far ones
    idx[::97, 0] = rng.integers(0, n, size=idx[::97, 0].shape)
    check(idx, n)
    check(idx, n, order=rng.permutation(n))


def test_transpose_padding_and_sources_in_another_order():
    rng = np.random.default_rng(1)
    m, n, ns = 3000, 2000, 24
    idx = rng.integers(0, n + 1, size=(m, ns))                             # n = the TF side's shadow index: left out
    idx[::7, 3] = -1
    check(idx, n)
    check(idx, n, order=rng.permutation(n), order_src=rng.permutation(m))   # the orders change the schedule, never the table


def test_transpose_hub_overflows_the_lds_stage():
    # every source lists target 5: one target tile receives 65536 pairs (> the 12288 an LDS stage holds) and orders them through global scratch
    m, n, ns = 4096, 4096, 16
    idx = np.full((m, ns), 5, np.int64)
    idx[:, 1] = np.arange(m)
    check(idx, n)


def test_transpose_of_a_real_search_with_its_cell_order():
    n, k = 40960, 16
    xyz, _ = S.s_room(n, seed=0)
    off = S.offsets(n, 1, 0)
    xyz_d, off_d = dev(xyz), dev(off)
    with pointops.neighbor_cache():
        idx, _ = pointops.knnquery_raw(k, xyz_d, xyz_d, off_d, off_d)
        order = pointops.spatial_order(idx)
        assert order is not None
        tr = pointops.neighbor_transpose(idx, n)
        assert tr is not None and tr[0] is order
        assert pointops.neighbor_transpose(idx, n)[1] is tr[1]             # second request: cache hit
        o, s, src = (t.cpu().numpy() for t in tr)
    rs, rsrc = transpose_numpy(idx.cpu().numpy(), n, o)
    assert np.array_equal(s, rs) and np.array_equal(src[:rs[-1]], rsrc)


@pytest.mark.parametrize("m,n,ka,kb,ordered", [(40960, 40960, 36, 16, True), (10240, 10240, 8, 36, True), (3000, 3000, 16, 16, False), (5000, 777, 3, 40, False),
                                                (70, 50, 65, 1, False), (140000, 140000, 4, 6, True)])
def test_two_tables_of_one_geometry_transposed_together(m, n, ka, kb, ordered):
    """cbl_neighbor_transpose_pair: both tables against the numpy restatement (the last case is beyond the pair launches' size: built one after the other inside the call)"""
    rng = np.random.default_rng(m + ka)
    idx_a = rng.integers(-1, n + 2, size=(m, ka)).astype(np.int32)       # shadow / padding entries on both sides of [0, n)
    idx_b = rng.integers(0, n, size=(m, kb)).astype(np.int32)
    idx_b[m // 2] = n // 3                                              # kb pairs of one source on one target
    order = rng.permutation(n).astype(np.int32) if ordered else None
    L = _lib.lib()
    a_d, b_d = dev(idx_a), dev(idx_b)
    o_d = None if order is None else dev(order)
    os_d = o_d if m == n else None
    outs = [torch.full((n + 1,), -7, dtype=torch.int32, device="cuda"), torch.full((m * ka,), -7, dtype=torch.int32, device="cuda"),
            torch.full((n + 1,), -7, dtype=torch.int32, device="cuda"), torch.full((m * kb,), -7, dtype=torch.int32, device="cuda")]
    L.cbl_neighbor_transpose_pair_workspace_bytes.restype = ctypes.c_size_t
    need = L.cbl_neighbor_transpose_pair_workspace_bytes(_i(m), _i(n), _i(ka), _i(kb))
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device="cuda")
    _lib.check(L.cbl_neighbor_transpose_pair(_i(m), _i(n), _i(ka), _lib.ptr(a_d), _i(kb), _lib.ptr(b_d), _lib.ptr(os_d), _lib.ptr(o_d), *[_lib.ptr(t) for t in outs],
                                             _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(a_d)), "cbl_neighbor_transpose_pair")
    torch.cuda.synchronize()
    for idx, st, src in ((idx_a, outs[0], outs[1]), (idx_b, outs[2], outs[3])):
        rs, rsrc = transpose_numpy(idx, n, order)
        assert np.array_equal(st.cpu().numpy(), rs)
        assert np.array_equal(src.cpu().numpy()[:rs[-1]], rsrc)


def test_a_companion_table_is_built_with_the_first_and_served_from_the_registry():
    """pointops.neighbor_transpose(companion=): the second table's later request launches nothing and returns what a build of its own returns"""
    xyz, _ = S.s_room(20000, 3)
    p = dev(xyz); o = torch.tensor([20000], dtype=torch.int32, device="cuda")
    with pointops.neighbor_cache() as nc:
        nc.hint(p, 36, "set")
        i16, _ = pointops.knnquery_raw(16, p, p, o, o)
        i36, _ = pointops.knnquery_raw(36, p, p, o, o, algo="set")
        order36, s36, src36 = pointops.neighbor_transpose(i36, 20000, companion=i16)
        hit = pointops.neighbor_transpose(i16, 20000, build=False)
        assert hit is not None and hit[0] is order36
        torch.cuda.synchronize()
        got = [t.cpu().numpy() for t in (s36, src36, hit[1], hit[2])]
        order = None if order36 is None else order36.cpu().numpy()
        a, b = i36.cpu().numpy(), i16.cpu().numpy()
    for idx, st, src in ((a, got[0], got[1]), (b, got[2], got[3])):
        rs, rsrc = transpose_numpy(idx, 20000, order)
        assert np.array_equal(st, rs) and np.array_equal(src[:rs[-1]], rsrc)


@pytest.mark.parametrize("c", [64, 32, 4, 3, 128, 20])
def test_grouping_backward_csr_bit_exact(c):
    rng = np.random.default_rng(c)
    m, n, ns = 6000, 5000, 16
    idx = rng.integers(0, n, size=(m, ns)).astype(np.int32)
    go = rng.normal(size=(m, ns, c)).astype(np.float32)
    ref = O.grouping_backward(go, idx, n)
    L = _lib.lib()
    for order in (None, rng.permutation(n).astype(np.int32)):
        s, src = transpose_numpy(idx, n, order)
        go_d, s_d, src_d = dev(go), dev(s), dev(src)
        o_d = None if order is None else dev(order)
        gi = torch.full((n, c), 3.0, dtype=torch.float32, device="cuda")   # written, not accumulated: no pre-zeroing needed
        _lib.check(L.cbl_grouping_backward_csr(_i(n), _i(c), _lib.ptr(go_d), _lib.ptr(o_d), _lib.ptr(s_d), _lib.ptr(src_d), _lib.ptr(gi),
                                               _lib.stream_of(go_d)), "cbl_grouping_backward_csr")
        assert np.array_equal(gi.cpu().numpy(), ref)


def test_autograd_grouping_backward_takes_the_gather_and_is_bit_exact():
    n, k, c = 40960, 16, 64
    xyz, _ = S.s_room(n, seed=2)
    off = S.offsets(n, 1, 2)
    rng = np.random.default_rng(2)
    feat = rng.normal(size=(n, c)).astype(np.float32)
    go = rng.normal(size=(n, k, c)).astype(np.float32)
    xyz_d, off_d = dev(xyz), dev(off)
    feat_d = dev(feat).requires_grad_(True)
    idx, _ = pointops.knnquery_raw(k, xyz_d, xyz_d, off_d, off_d)
    out = pointops.grouping(feat_d, idx)
    out.backward(dev(go))
    ref = O.grouping_backward(go, idx.cpu().numpy(), n)
    assert np.array_equal(feat_d.grad.cpu().numpy(), ref)
    # run-to-run deterministic (the atomic scatter was not)
    feat_d.grad = None
    pointops.grouping(feat_d, idx).backward(dev(go))
    assert np.array_equal(feat_d.grad.cpu().numpy(), ref)


def test_contrast_gradient_is_deterministic_and_matches_the_atomic_kernels():
    from contrastboundary_amd import heads
    n, k, d = 20000, 36, 32
    xyz, labels = S.s_room(n, seed=5)
    off = S.offsets(n, 2, 5)
    rng = np.random.default_rng(5)
    latent = rng.normal(size=(n, d)).astype(np.float32)
    xyz_d, off_d, lab_d = dev(xyz), dev(off), dev(labels)
    idx, _ = pointops.knnquery_raw(k, xyz_d, xyz_d, off_d, off_d, algo="set")
    grads, losses = [], []
    for _ in range(2):
        f = dev(latent).requires_grad_(True)
        loss = heads.point_contrast(f, lab_d, idx, 1.0, 0.1)
        loss.backward()
        grads.append(f.grad.cpu().numpy()); losses.append(loss.item())
    assert losses[0] == losses[1] and np.array_equal(grads[0], grads[1])
    # round 1's fused forward + gradient with atomics (still exported): same numbers up to summation order
    L = _lib.lib()
    f = dev(latent)
    per_point = torch.empty(n, device="cuda"); mask = torch.empty(n, dtype=torch.int32, device="cuda")
    stats = torch.empty(2, device="cuda"); loss_o = torch.empty(1, device="cuda"); unit = torch.zeros_like(f)
    amax = lab_d.to(torch.int32)
    _lib.check(L.cbl_point_contrast_forward_grad(_i(n), _i(k), _i(d), _lib.ptr(f), _lib.ptr(amax), _lib.ptr(idx), ctypes.c_float(1.0), ctypes.c_float(0.1),
                                                 _lib.ptr(per_point), _lib.ptr(mask), _lib.ptr(stats), _lib.ptr(loss_o), _lib.ptr(unit), _lib.stream_of(f)), "fwd_grad")
    old = (unit * (0.1 / stats[1])).cpu().numpy()
    assert abs(loss_o.item() - losses[0]) < 1e-6 * max(1.0, abs(losses[0]))
    assert np.allclose(grads[0], old, rtol=1e-4, atol=1e-7 + 1e-4 * np.abs(old).max())


# ---- the other scatter-adds of the path as gathers over the same table: K6, K8, K10 through their autograd functions, above the size at
# which the table is built.  Pairs are summed in ascending order = the order of the reference loops run sequentially: bit-exact, deterministic.
def _scene(n, k, seed):
    xyz, _ = S.s_room(n, seed=seed)
    off = S.offsets(n, 2, seed)
    xyz_d, off_d = dev(xyz), dev(off)
    idx, _ = pointops.knnquery_raw(k, xyz_d, xyz_d, off_d, off_d)
    return xyz_d, off_d, idx


def test_subtraction_backward_as_a_gather_bit_exact():
    n, k, c = 12000, 16, 32
    _, _, idx = _scene(n, k, 7)
    rng = np.random.default_rng(7)
    go = rng.normal(size=(n, k, c)).astype(np.float32)
    a = dev(rng.normal(size=(n, c)).astype(np.float32)).requires_grad_(True)
    b = dev(rng.normal(size=(n, c)).astype(np.float32)).requires_grad_(True)
    assert n * k >= pointops.TRANSPOSE_MIN_PAIRS
    res = []
    for _ in range(2):
        a.grad = b.grad = None
        pointops.subtraction(a, b, idx).backward(dev(go))
        res.append((a.grad.cpu().numpy(), b.grad.cpu().numpy()))
    g1, g2 = O.subtraction_backward(idx.cpu().numpy(), go)
    assert np.array_equal(res[0][0], g1) and np.array_equal(res[0][1], g2)
    assert np.array_equal(res[0][1], res[1][1])
    assert pointops.neighbor_transpose(idx, n, build=False) is not None           # the gather path ran (and left its table)


@pytest.mark.parametrize("c,wc", [(32, 4), (64, 8), (24, 3)])
def test_aggregation_backward_as_a_gather_bit_exact(c, wc):
    n, k = 9000, 16
    _, _, idx = _scene(n, k, 8)
    rng = np.random.default_rng(c)
    x = rng.normal(size=(n, c)).astype(np.float32); pos = rng.normal(size=(n, k, c)).astype(np.float32)
    w = rng.normal(size=(n, k, wc)).astype(np.float32); go = rng.normal(size=(n, c)).astype(np.float32)
    xd, pd, wd = (dev(t).requires_grad_(True) for t in (x, pos, w))
    pointops.aggregation(xd, pd, wd, idx).backward(dev(go))
    gi, gp, gw = O.aggregation_backward(x, pos, w, idx.cpu().numpy(), go)
    assert np.array_equal(xd.grad.cpu().numpy(), gi)                             # ascending pairs: the sequential loop's sums
    assert np.array_equal(pd.grad.cpu().numpy(), gp)
    assert np.allclose(wd.grad.cpu().numpy(), gw, rtol=1e-4, atol=1e-4 * np.abs(gw).max())     # in-wave reduction over the channels sharing a weight
    assert pointops.neighbor_transpose(idx, n, build=False) is not None


def test_interpolation_backward_as_a_gather_bit_exact():
    n_coarse, n_fine, c = 8000, 32000, 32
    xyz, _ = S.s_room(n_fine, seed=9)
    rng = np.random.default_rng(9)
    sel = np.sort(rng.choice(n_fine, n_coarse, replace=False))
    fine, coarse = dev(xyz), dev(xyz[sel])
    of, oc = dev(np.array([n_fine], np.int32)), dev(np.array([n_coarse], np.int32))
    idx, dist = pointops.knnquery_raw(3, coarse, fine, oc, of)
    d = torch.sqrt(dist)
    wgt = (1.0 / (d + 1e-8)); wgt = (wgt / wgt.sum(1, keepdim=True)).contiguous()
    feat = dev(rng.normal(size=(n_coarse, c)).astype(np.float32)).requires_grad_(True)
    go = rng.normal(size=(n_fine, c)).astype(np.float32)
    assert n_fine * 3 >= pointops.TRANSPOSE_MIN_PAIRS
    # through the kernel pair directly (the autograd function computes idx / weight itself)
    L = _lib.lib()
    tr = pointops.neighbor_transpose(idx, n_coarse)
    assert tr is not None
    order, inv_start, inv_src = tr
    gi = torch.full((n_coarse, c), 5.0, device="cuda")
    go_d = dev(go)
    _lib.check(L.cbl_weighted_scatter_csr(_i(n_coarse), _i(3), _i(c), _i(1), _lib.ptr(go_d), _lib.ptr(wgt), _lib.ptr(order), _lib.ptr(inv_start),
                                          _lib.ptr(inv_src), _lib.ptr(gi), _lib.stream_of(go_d)), "cbl_weighted_scatter_csr")
    ref = O.interpolation_backward(go, idx.cpu().numpy(), wgt.cpu().numpy(), n_coarse)
    assert np.array_equal(gi.cpu().numpy(), ref)
    # and through autograd: same numbers as the atomic kernel up to summation order, deterministic
    grads = []
    for _ in range(2):
        feat.grad = None
        pointops.interpolation2(coarse, fine, feat, oc, of, 3).backward(go_d)
        grads.append(feat.grad.cpu().numpy())
    assert np.array_equal(grads[0], grads[1])
