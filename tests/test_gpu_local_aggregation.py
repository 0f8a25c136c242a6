"""GPU parity: KPConv (MFMA) / AdaptiveWeight / index pooling vs the numpy restatement of the TF graph code
(oracle/local_aggregation_oracle.py; parity unpinned by execution — TF absent).  Tolerance 1e-4 (north_star)."""
import numpy as np
import pytest
import torch

from oracle import local_aggregation_oracle as LA

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make(n0, n, K, C, seed, pad_frac=0.2):
    rng = np.random.default_rng(seed)
    s = rng.uniform(0, 1, (n0, 3)).astype(np.float32)
    q = s[rng.choice(n0, n, replace=False)] + rng.normal(0, 0.01, (n, 3)).astype(np.float32)
    idx = rng.integers(0, n0, (n, K)).astype(np.int32)
    # neighbours close to the query so that the linear influence is non-trivial + trailing shadow (== n0) padding like the radius search
    d = ((s[None, :, :] - q[:, None, :]) ** 2).sum(-1)
    idx = np.argsort(d, 1)[:, :K].astype(np.int32)
    npad = rng.integers(0, int(K * pad_frac) + 1, n)
    for i in range(n):
        if npad[i]:
            idx[i, K - npad[i]:] = n0
    f = rng.normal(size=(n0, C)).astype(np.float32)
    return q.astype(np.float32), s, idx, f, rng


@pytest.mark.parametrize("K,C,KP,influence,mode", [(16, 64, 15, "linear", "sum"), (26, 72, 15, "linear", "sum"), (9, 16, 7, "linear", "closest"),
                                                   (31, 144, 15, "constant", "sum"), (5, 20, 16, "linear", "sum"),
                                                   # the C <= 64 kernel beyond one chunk of 16 neighbours, with a constant influence, beyond 64 neighbours
                                                   (26, 64, 15, "linear", "sum"), (40, 32, 15, "constant", "sum"), (70, 48, 13, "linear", "closest"),
                                                   (100, 64, 15, "linear", "sum"), (16, 8, 15, "constant", "closest")])
def test_kpconv(K, C, KP, influence, mode):
    from contrastboundary_amd import local_aggregation as L
    q, s, idx, f, rng = make(700, 300, K, C, seed=K)
    kpts = (rng.normal(size=(KP, 3)) * 0.06).astype(np.float32); kpts[0] = 0
    kw = rng.normal(size=(KP, C)).astype(np.float32)
    extent = 0.09
    ft = dev(f).requires_grad_(True); kwt = dev(kw).requires_grad_(True)
    out = L.kpconv(dev(q), dev(s), dev(idx), ft, dev(kpts), kwt, extent, influence, mode)
    ref = LA.kpconv(q, s, idx, f, kpts, kw, extent, influence, mode)
    scale = np.abs(ref).max()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * scale)
    go = rng.normal(size=ref.shape).astype(np.float32)
    if K > 64:                                                       # the backward kernels take K <= 64 (the reference's K_lim is <= 41): loud, not wrong
        from contrastboundary_amd._lib import CblError
        with pytest.raises(CblError):
            out.backward(dev(go))
        return
    out.backward(dev(go))
    gf, gkw = LA.kpconv_grads(q, s, idx, f, kpts, kw, extent, go, influence, mode)
    np.testing.assert_allclose(ft.grad.cpu().numpy(), gf, rtol=1e-4, atol=1e-4 * np.abs(gf).max())
    np.testing.assert_allclose(kwt.grad.cpu().numpy(), gkw, rtol=1e-4, atol=1e-4 * np.abs(gkw).max())


@pytest.mark.parametrize("K,C,reduction", [(26, 72, "mean"), (16, 64, "mean"), (41, 100, "sum")])
def test_adaptive_weight(K, C, reduction):
    from contrastboundary_amd import local_aggregation as L
    q, s, idx, f, rng = make(900, 400, K, C, seed=C)
    W = (rng.normal(size=(3, C)) * 0.5).astype(np.float32); b = rng.normal(size=(C,)).astype(np.float32)
    radius = 0.1
    ft = dev(f).requires_grad_(True); Wt = dev(W).requires_grad_(True); bt = dev(b).requires_grad_(True)
    out = L.adaptive_weight(dev(q), dev(s), dev(idx), ft, radius, Wt, bt, reduction)
    ref = LA.adaptive_weight(q, s, idx, f, radius, W, b, reduction)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    go = rng.normal(size=ref.shape).astype(np.float32)
    out.backward(dev(go))
    gf, gW, gb = LA.adaptive_weight_grads(q, s, idx, f, radius, W, b, go, reduction)
    np.testing.assert_allclose(ft.grad.cpu().numpy(), gf, rtol=1e-4, atol=1e-4 * np.abs(gf).max())
    np.testing.assert_allclose(Wt.grad.cpu().numpy(), gW, rtol=1e-4, atol=1e-4 * np.abs(gW).max())
    np.testing.assert_allclose(bt.grad.cpu().numpy(), gb, rtol=1e-4, atol=1e-4 * np.abs(gb).max())


@pytest.mark.parametrize("C,K,coherent,reduction", [(72, 26, True, "mean"), (72, 26, False, "mean"), (144, 31, True, "mean"), (288, 38, True, "sum"), (64, 20, True, "mean"),
                                                    (1152, 39, True, "mean"), (72, 64, False, "mean")])
def test_adaptive_weight_forward_at_pyramid_widths(C, K, coherent, reduction):
    """the forward at the ConvNet's widths (lanes own 12 channels where C/4 divides by 3, else 8: adaptive_weight_fwd_v5; the fully connected layer
    factored out of the neighbour loop), queries in cell order and in random order, shadow padding, a partial last trip, K up to 64"""
    from contrastboundary_amd import local_aggregation as L
    n0, n = 5000, 4500 + 5
    rng = np.random.default_rng(C + K)
    s = rng.uniform(0, 1, (n0, 3)).astype(np.float32)
    q = s[rng.choice(n0, n, replace=False)] + rng.normal(0, 0.01, (n, 3)).astype(np.float32)
    if coherent:
        cell = np.floor(q / 0.12).astype(np.int64)
        q = q[np.lexsort((cell[:, 2], cell[:, 1], cell[:, 0]))]
    idx = np.empty((n, K), np.int32)
    for a in range(0, n, 500):
        d = ((s[None, :, :] - q[a:a + 500, None, :]) ** 2).sum(-1)
        idx[a:a + 500] = np.argsort(d, 1)[:, :K]
    npad = rng.integers(0, K // 4 + 1, n)
    for i in range(n):
        if npad[i]:
            idx[i, K - npad[i]:] = n0
    f = rng.normal(size=(n0, C)).astype(np.float32)
    W = (rng.normal(size=(3, C)) * 0.5).astype(np.float32); b = rng.normal(size=(C,)).astype(np.float32)
    out = L.adaptive_weight(dev(q), dev(s), dev(idx), dev(f), 0.1, dev(W), dev(b), reduction)
    ref = LA.adaptive_weight(q, s, idx, f, 0.1, W, b, reduction)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    out2 = L.adaptive_weight(dev(q), dev(s), dev(idx), dev(f), 0.1, dev(W), dev(b), reduction)
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), out2.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("n0,n,K,C,reduction", [(3000, 2600, 26, 72, "mean"), (2000, 2000, 31, 144, "mean"), (900, 700, 38, 288, "sum"),
                                                 (500, 400, 41, 576, "mean"), (300, 260, 39, 1152, "mean"), (800, 800, 9, 8, "mean")])
def test_adaptive_weight_backward_as_a_gather(n0, n, K, C, reduction):
    """the backward over the transposed neighbour table (no atomics): every lane / chunk geometry of the kernel (C/4 = 18, 36, 72, 144, 2 x 144, 2
    lanes per target), query set != support set, shadow padding; values against the restatement, run-to-run bitwise identical"""
    from contrastboundary_amd import local_aggregation as L, pointops
    q, s, idx, f, rng = make(n0, n, K, C, seed=C + K)
    W = (rng.normal(size=(3, C)) * 0.5).astype(np.float32); b = rng.normal(size=(C,)).astype(np.float32)
    radius = 0.1
    go = rng.normal(size=(n, C)).astype(np.float32)
    grads = []
    idx_d = dev(idx)
    assert pointops.neighbor_transpose(idx_d, n0) is not None       # the table exists: the backward takes the gather path whatever the size
    for _ in range(2):
        ft = dev(f).requires_grad_(True); Wt = dev(W).requires_grad_(True); bt = dev(b).requires_grad_(True)
        out = L.adaptive_weight(dev(q), dev(s), idx_d, ft, radius, Wt, bt, reduction)
        out.backward(dev(go))
        grads.append((ft.grad.cpu().numpy(), Wt.grad.cpu().numpy(), bt.grad.cpu().numpy()))
    gf, gW, gb = LA.adaptive_weight_grads(q, s, idx, f, radius, W, b, go, reduction)
    np.testing.assert_allclose(grads[0][0], gf, rtol=1e-4, atol=1e-4 * np.abs(gf).max())
    np.testing.assert_allclose(grads[0][1], gW, rtol=1e-4, atol=1e-4 * np.abs(gW).max())
    np.testing.assert_allclose(grads[0][2], gb, rtol=1e-4, atol=1e-4 * np.abs(gb).max())
    for a, c in zip(grads[0], grads[1]):
        np.testing.assert_array_equal(a.view(np.uint32), c.view(np.uint32))      # no atomics: deterministic
    # only the feature gradient / only the parameter gradients
    ft = dev(f).requires_grad_(True)
    L.adaptive_weight(dev(q), dev(s), idx_d, ft, radius, dev(W), dev(b), reduction).backward(dev(go))
    np.testing.assert_array_equal(ft.grad.cpu().numpy().view(np.uint32), grads[0][0].view(np.uint32))
    Wt = dev(W).requires_grad_(True); bt = dev(b).requires_grad_(True)
    L.adaptive_weight(dev(q), dev(s), idx_d, dev(f), radius, Wt, bt, reduction).backward(dev(go))
    np.testing.assert_array_equal(Wt.grad.cpu().numpy().view(np.uint32), grads[0][1].view(np.uint32))
    np.testing.assert_array_equal(bt.grad.cpu().numpy().view(np.uint32), grads[0][2].view(np.uint32))


def test_adaptive_weight_mean_quirk_without_padding():
    # no row is padded -> padding_num = max(idx) is a REAL index and rows containing it count one neighbour less (:466-470)
    from contrastboundary_amd import local_aggregation as L
    q, s, idx, f, rng = make(300, 100, 8, 16, seed=1, pad_frac=0.0)
    W = rng.normal(size=(3, 16)).astype(np.float32); b = rng.normal(size=(16,)).astype(np.float32)
    out = L.adaptive_weight(dev(q), dev(s), dev(idx), dev(f), 0.2, dev(W), dev(b), "mean")
    np.testing.assert_allclose(out.cpu().numpy(), LA.adaptive_weight(q, s, idx, f, 0.2, W, b, "mean"), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n1,n2,k,d", [(500, 200, 9, 37), (5000, 3000, 26, 72), (900, 400, 31, 1152), (700, 700, 5, 8), (3000, 1, 41, 144)])
def test_index_pooling(n1, n2, k, d):
    """ind_max_pool / ind_closest_pool (basic_operators.py:155-192), bit for bit: the one-channel-per-lane kernel (d = 37) and the float4 kernel with its
    column chunks (d = 1152: two chunks of 144 lanes), rows that are all shadow, k not a multiple of the rows in flight"""
    from contrastboundary_amd import local_aggregation as L
    rng = np.random.default_rng(n1 + d)
    x = rng.normal(size=(n1, d)).astype(np.float32)
    inds = rng.integers(0, n1 + 1, (n2, k)).astype(np.int32)          # n1 == shadow
    inds[0, :] = n1
    if n2 > 3:
        inds[3, 1:] = n1
    np.testing.assert_array_equal(L.ind_max_pool(dev(x), dev(inds)).cpu().numpy(), LA.ind_max_pool(x, inds))
    np.testing.assert_array_equal(L.ind_closest_pool(dev(x), dev(inds)).cpu().numpy(), LA.ind_closest_pool(x, inds))


def test_kpconv_full_size_linearity():
    """C2 size: KPConv is linear in the features and in the kernel weights"""
    from contrastboundary_amd import hotpath, local_aggregation as L, pointops
    sc = hotpath.Scene.synthetic(40960, 64, seed=0)
    idx, _ = pointops.knnquery_raw(16, sc.xyz, sc.xyz, sc.offset, sc.offset)
    g = torch.Generator(device="cuda").manual_seed(0)
    kpts = torch.randn(15, 3, device="cuda", generator=g) * 0.06
    kw = torch.randn(15, 64, device="cuda", generator=g)
    f2 = torch.randn(40960, 64, device="cuda", generator=g)
    a = L.kpconv(sc.xyz, sc.xyz, idx, sc.feat, kpts, kw, 0.12)
    b = L.kpconv(sc.xyz, sc.xyz, idx, f2, kpts, kw, 0.12)
    ab = L.kpconv(sc.xyz, sc.xyz, idx, sc.feat + 2 * f2, kpts, kw, 0.12)
    assert torch.allclose(ab, a + 2 * b, rtol=1e-4, atol=1e-3)
    assert torch.isfinite(a).all() and float(a.abs().max()) > 0


@pytest.mark.parametrize("pe,C,reduction,K", [("sin_cos", 72, "mean", 26), ("sin_cos", 9, "mean", 16), ("sin_cos", 144, "sum", 31), ("sin_cos", 36, "max", 20),
                                              ("xyz", 72, "mean", 26), ("one", 40, "sum", 12), ("distance", 33, "mean", 18), ("exp_-d", 64, "max", 26),
                                              ("direction_exp_-d", 72, "mean", 26), ("direction_exp_-d", 18, "sum", 9), ("direction_d", 9, "max", 17),
                                              ("direction_d", 64, "mean", 38), ("two_order", 72, "mean", 26), ("three_order", 9, "sum", 26),
                                              ("three_order", 144, "mean", 41)])
def test_pospool(pe, C, reduction, K):
    """a14 PosPool: every runnable position embedding x reduction vs the numpy restatement (forward 1e-4, feature gradient 1e-3 rel)"""
    from contrastboundary_amd import local_aggregation as L
    q, s, idx, f, rng = make(800, 350, K, C, seed=K + C)
    radius = 0.1
    ft = dev(f).requires_grad_(True)
    out = L.pospool(dev(q), dev(s), dev(idx), ft, radius, pe, reduction)
    ref, _, _ = LA.pospool(q, s, idx, f, radius, pe, reduction)
    scale = max(np.abs(ref).max(), 1.0)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * scale)
    go = rng.normal(size=ref.shape).astype(np.float32)
    out.backward(dev(go))
    gf = LA.pospool_grad_features(q, s, idx, f, radius, go, pe, reduction)
    np.testing.assert_allclose(ft.grad.cpu().numpy(), gf, rtol=1e-4, atol=1e-4 * max(np.abs(gf).max(), 1.0))


@pytest.mark.parametrize("n0,n,K,C,pe,reduction", [(3000, 2600, 26, 72, "sin_cos", "mean"), (2000, 2000, 31, 144, "sin_cos", "sum"), (900, 700, 38, 288, "xyz", "mean"),
                                                    (300, 260, 39, 1152, "sin_cos", "mean"), (800, 800, 9, 8, "one", "mean"), (600, 600, 12, 64, "direction_d", "mean"), (700, 700, 41, 72, "three_order", "sum")])
def test_pospool_backward_as_a_gather(n0, n, K, C, pe, reduction):
    """'sum' / 'mean' backward over the transposed neighbour table (no atomics): the lane / chunk geometries of the kernel, query set != support set,
    shadow padding; values against the restatement, bitwise identical from run to run, and equal (1e-4) to the scatter entry it replaces"""
    import ctypes
    from contrastboundary_amd import _lib, local_aggregation as L, pointops
    q, s, idx, f, rng = make(n0, n, K, C, seed=C + K)
    radius = 0.1
    go = rng.normal(size=(n, C)).astype(np.float32)
    idx_d = dev(idx)
    assert pointops.neighbor_transpose(idx_d, n0) is not None       # the table exists: the backward takes the gather path whatever the size
    grads = []
    for _ in range(2):
        ft = dev(f).requires_grad_(True)
        L.pospool(dev(q), dev(s), idx_d, ft, radius, pe, reduction).backward(dev(go))
        grads.append(ft.grad.cpu().numpy())
    gf = LA.pospool_grad_features(q, s, idx, f, radius, go, pe, reduction)
    np.testing.assert_allclose(grads[0], gf, rtol=1e-4, atol=1e-4 * max(np.abs(gf).max(), 1.0))
    np.testing.assert_array_equal(grads[0].view(np.uint32), grads[1].view(np.uint32))
    # the scatter entry (float atomics) on the same inputs
    lib = _lib.lib()
    i, fl = ctypes.c_int, ctypes.c_float
    pad = torch.empty(1, dtype=torch.int32, device="cuda")
    _lib.check(lib.cbl_index_max(ctypes.c_longlong(n * K), _lib.ptr(idx_d), _lib.ptr(pad), _lib.stream_of(idx_d)), "cbl_index_max")
    qd, sd, fd, god = dev(q), dev(s), dev(f), dev(go)
    sc = torch.zeros(n0, C, device="cuda")
    _lib.check(lib.cbl_pospool_backward(i(n), i(n0), i(K), i(C), _lib.ptr(qd), _lib.ptr(sd), _lib.ptr(idx_d), _lib.ptr(fd), fl(radius),
                                        i(L.POSPOOL_EMBEDDINGS[pe]), i({"sum": 0, "mean": 1}[reduction]), _lib.ptr(pad), _lib.ptr(god), _lib.ptr(sc),
                                        _lib.stream_of(fd)), "cbl_pospool_backward")
    np.testing.assert_allclose(grads[0], sc.cpu().numpy(), rtol=1e-4, atol=1e-4 * max(np.abs(gf).max(), 1.0))


def test_pospool_rejects_what_the_reference_cannot_reshape():
    from contrastboundary_amd import local_aggregation as L
    q, s, idx, f, rng = make(100, 50, 8, 10, seed=1)
    with pytest.raises(RuntimeError):
        L.pospool(dev(q), dev(s), dev(idx), dev(f), 0.1, "sin_cos", "mean")       # 10 is neither 9 nor a multiple of 6
    with pytest.raises(NotImplementedError):
        L.pospool(dev(q), dev(s), dev(idx), dev(f), 0.1, "direction", "mean")


@pytest.mark.parametrize("K,C,KP,influence,mode", [(16, 64, 15, "linear", "sum"), (26, 72, 15, "linear", "sum"), (9, 16, 7, "linear", "closest"),
                                                   (31, 144, 15, "constant", "sum"), (5, 20, 16, "linear", "sum")])
def test_kpconv_backward_as_a_gather(K, C, KP, influence, mode):
    """cbl_kpconv_backward_csr (transposed neighbour table, no atomics) against the numpy restatement's analytic gradients, 1e-4 of their
    scale; also: either gradient alone, the shadow padding of the radius search, run-to-run identical bits"""
    import ctypes
    from contrastboundary_amd import _lib, pointops
    q, s, idx, f, rng = make(700, 300, K, C, seed=K + 1)
    kpts = (rng.normal(size=(KP, 3)) * 0.06).astype(np.float32); kpts[0] = 0
    kw = rng.normal(size=(KP, C)).astype(np.float32)
    extent = 0.09
    go = rng.normal(size=(300, C)).astype(np.float32)
    gf_ref, gkw_ref = LA.kpconv_grads(q, s, idx, f, kpts, kw, extent, go, influence, mode)
    L = _lib.lib()
    i = ctypes.c_int
    idx_d = dev(idx)
    tr = pointops.neighbor_transpose(idx_d, 700)
    assert tr is not None
    order, inv_start, inv_src = tr
    qd, sd, fd, kpd, kwd, god = dev(q), dev(s), dev(f), dev(kpts), dev(kw), dev(go)
    ws = torch.empty(max(L.cbl_kpconv_backward_csr_workspace_bytes(i(700), i(C), i(KP)), 1), dtype=torch.uint8, device="cuda")
    outs = []
    for want_f, want_w in ((True, True), (True, False), (False, True), (True, True)):
        gf = torch.full((700, C), 7.0, device="cuda") if want_f else None      # written, not accumulated
        gkw = torch.full((KP, C), 7.0, device="cuda") if want_w else None
        _lib.check(L.cbl_kpconv_backward_csr(i(300), i(700), i(K), i(C), i(KP), _lib.ptr(qd), _lib.ptr(sd), _lib.ptr(fd), _lib.ptr(kpd), _lib.ptr(kwd),
                                             ctypes.c_float(extent), i(1 if influence == "linear" else 0), i(1 if mode == "closest" else 0), _lib.ptr(god),
                                             _lib.ptr(order), _lib.ptr(inv_start), _lib.ptr(inv_src), _lib.ptr(gf), _lib.ptr(gkw), _lib.ptr(ws),
                                             ctypes.c_size_t(ws.numel()), _lib.stream_of(fd)), "cbl_kpconv_backward_csr")
        if want_f:
            np.testing.assert_allclose(gf.cpu().numpy(), gf_ref, rtol=1e-4, atol=1e-4 * np.abs(gf_ref).max())
        if want_w:
            np.testing.assert_allclose(gkw.cpu().numpy(), gkw_ref, rtol=1e-4, atol=1e-4 * np.abs(gkw_ref).max())
        outs.append((gf, gkw))
    assert torch.equal(outs[0][0], outs[3][0]) and torch.equal(outs[0][1], outs[3][1])


def test_kpconv_shadow_neighbours_do_not_read_row_zero():
    """a neighbour that is not real (shadow index n0) contributes the reference's zero feature row (local_aggregation_operators.py:713) whatever the
    table holds: the kernels load from a clamped row 0 unconditionally, and a non-finite value there must not leak as 0 * inf"""
    from contrastboundary_amd import local_aggregation as L
    q, s, idx, f, rng = make(600, 300, 12, 32, seed=3)
    idx = idx.copy(); idx[::3, -4:] = 300                              # shadow neighbours on every third point
    kpts = (rng.normal(size=(15, 3)) * 0.06).astype(np.float32); kpts[0] = 0
    kw = rng.normal(size=(15, 32)).astype(np.float32)
    ref = LA.kpconv(q, s, idx, f, kpts, kw, 0.09, "constant", "sum")
    f_bad = f.copy(); f_bad[0] = np.inf
    used0 = (idx == 0).any(1)                                          # points that really list row 0 become non-finite, nobody else
    for influence in ("constant", "linear"):
        out = L.kpconv(dev(q), dev(s), dev(idx), dev(f_bad), dev(kpts), dev(kw), 0.09, influence, "sum").cpu().numpy()
        assert np.isfinite(out[~used0]).all()
    out = L.kpconv(dev(q), dev(s), dev(idx), dev(f_bad), dev(kpts), dev(kw), 0.09, "constant", "sum").cpu().numpy()
    np.testing.assert_allclose(out[~used0], ref[~used0], rtol=1e-4, atol=1e-4 * np.abs(ref).max())
