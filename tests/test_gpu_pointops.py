"""GPU parity: HIP kernels (through the C ABI, via contrastboundary_amd.pointops) vs the CPU oracle and the
golden vectors from the reference kernel bodies.  idx / integer outputs bit-exact; forward float outputs
bit-exact (same rounding, no FMA); atomically-accumulated gradients within 1e-4 (north_star tolerance)."""
import os

import numpy as np
import pytest
import torch

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
ATOL = 1e-4   # BASELINE.json north_star: "float outputs within 1e-4"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def names(npz, prefix):
    return sorted({k.split("/")[1] for k in npz.files if k.startswith(prefix + "/")})


KNN = np.load(os.path.join(G, "pointops_knn.npz"))
FPS = np.load(os.path.join(G, "pointops_fps.npz"))
K310 = np.load(os.path.join(G, "pointops_k3_k10.npz"))


@pytest.fixture(scope="module")
def P():
    from contrastboundary_amd import pointops
    return pointops


def test_library_is_native_gfx950():
    from contrastboundary_amd import _lib
    assert _lib.lib().cbl_device_arch_ok() == 1


@pytest.mark.parametrize("algo", ["exact", "auto"])
@pytest.mark.parametrize("name", names(KNN, "knn"))
def test_knn_golden(P, name, algo):
    g = lambda f: KNN[f"knn/{name}/{f}"]
    idx, d2 = P.knnquery_raw(int(g("k")), dev(g("xyz")), dev(g("new_xyz")), dev(g("offset")), dev(g("new_offset")), algo=algo)
    np.testing.assert_array_equal(idx.cpu().numpy(), g("idx"))
    np.testing.assert_array_equal(d2.cpu().numpy().view(np.uint32), g("dist2").view(np.uint32))


@pytest.mark.parametrize("algo", ["exact", "auto"])
@pytest.mark.parametrize("n,m,b,k,seed", [(5000, 5000, 1, 16, 0), (6000, 1500, 3, 16, 1), (3000, 3000, 4, 36, 2),
                                          (2000, 700, 2, 3, 3), (2000, 2000, 1, 1, 4), (4096, 64, 2, 256, 5),
                                          (1500, 200, 1, 400, 6), (8000, 2000, 2, 24, 7), (9000, 9000, 1, 64, 8), (5000, 5000, 1, 33, 9),
                                          (6000, 900, 3, 48, 10), (20000, 300, 1, 256, 11), (12000, 500, 2, 100, 12), (9000, 64, 1, 1000, 13),
                                          (40960, 160, 1, 256, 14)])
def test_knn_vs_oracle_random(P, n, m, b, k, seed, algo):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(0, 2, (n, 3)).astype(np.float32)
    cuts = np.sort(rng.choice(np.arange(1, n), b - 1, replace=False)) if b > 1 else np.array([], int)
    offset = np.concatenate([cuts, [n]]).astype(np.int32)
    if m == n:
        q, noff = xyz, offset
    else:   # queries: a random subset per cloud, jittered
        lens = np.diff(np.concatenate([[0], offset]))
        take = np.maximum(1, (lens * m // n)).astype(int)
        sel = np.concatenate([np.sort(rng.choice(l, t, replace=False)) + s for l, t, s in zip(lens, take, np.concatenate([[0], offset[:-1]]))])
        q = (xyz[sel] + rng.normal(0, 0.01, (len(sel), 3))).astype(np.float32)
        noff = np.cumsum(take).astype(np.int32)
    idx, d2 = P.knnquery_raw(k, dev(xyz), dev(q), dev(offset), dev(noff), algo=algo)
    ridx, rd2 = O.knnquery(k, xyz, q, offset, noff)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(d2.cpu().numpy().view(np.uint32), rd2.view(np.uint32))


def test_knn_python_api_matches_reference_signature(P):
    # knnquery(nsample, xyz, new_xyz=None, offset, new_offset) -> (idx int32, dist = sqrt(dist2)); nsample may be a 0-dim tensor
    rng = np.random.default_rng(0)
    xyz = dev(rng.uniform(size=(300, 3)).astype(np.float32))
    o = dev(np.int32([300]))
    idx, dist = P.knnquery(torch.tensor(5), xyz, None, o, o)
    assert idx.dtype == torch.int32 and idx.shape == (300, 5) and dist.shape == (300, 5)
    ridx, rd2 = O.knnquery(5, xyz.cpu().numpy(), xyz.cpu().numpy(), [300], [300])
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(dist.cpu().numpy(), np.sqrt(rd2))
    with pytest.raises(Exception):
        P.knnquery(5, xyz.cpu(), None, o, o)          # CPU tensors are rejected loudly, no fallback


@pytest.mark.parametrize("name", names(FPS, "fps"))
def test_fps_golden(P, name):
    g = lambda f: FPS[f"fps/{name}/{f}"]
    idx = P.furthestsampling(dev(g("xyz")), dev(g("offset")), dev(g("new_offset")))
    np.testing.assert_array_equal(idx.cpu().numpy(), g("idx"))


@pytest.mark.parametrize("n,b,seed", [(900, 1, 0), (3000, 2, 1), (9000, 1, 2), (14000, 1, 3), (30000, 2, 4), (45000, 1, 5)])
def test_fps_vs_oracle(P, n, b, seed):
    # through the mirror: the dense kernels below 3072 points per cloud, the bucket-pruned kernel above
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(0, 3, (n, 3)).astype(np.float32)
    cuts = np.sort(rng.choice(np.arange(100, n - 100), b - 1, replace=False)) if b > 1 else np.array([], int)
    offset = np.concatenate([cuts, [n]]).astype(np.int32)
    lens = np.diff(np.concatenate([[0], offset]))
    noff = np.cumsum(np.minimum(lens // 4, 400)).astype(np.int32)       # <= 400 samples per cloud keeps the oracle fast
    idx = P.furthestsampling(dev(xyz), dev(offset), dev(noff))
    ridx, _ = O.furthestsampling(xyz, offset, noff)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)


def test_fps_lattice_ties_all_block_sizes(P):
    # lattices make every distance tie; n_max selects the reference block size 2^floor(log2 n_max)
    for side in (3, 5, 7, 11):
        g = np.arange(side, dtype=np.float32)
        xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        n = xyz.shape[0]
        idx = P.furthestsampling(dev(xyz), dev(np.int32([n])), dev(np.int32([n // 3])))
        ridx, _ = O.furthestsampling(xyz, [n], [n // 3])
        np.testing.assert_array_equal(idx.cpu().numpy(), ridx)


def _fps_raw(xyz, offset, new_offset, bucket):
    """direct C-ABI call: the dense kernels (cbl_furthestsampling) or the bucket-pruned one (cbl_furthestsampling_ws); -> idx, tmp"""
    import ctypes
    from contrastboundary_amd import _lib
    L = _lib.lib()
    x, o, no = dev(xyz), dev(np.int32(offset)), dev(np.int32(new_offset))
    n, b = x.shape[0], o.shape[0]
    n_max = int(np.max(np.diff(np.concatenate([[0], np.int32(offset)]))))
    idx = torch.zeros(int(new_offset[-1]), dtype=torch.int32, device="cuda")
    tmp = torch.full((n,), 1e10, dtype=torch.float32, device="cuda")
    st = _lib.stream_of(x)
    if bucket:
        need = L.cbl_furthestsampling_workspace_bytes(b, n, n_max)
        assert need > 0, "bucket kernel not selected for this size"
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        _lib.check(L.cbl_furthestsampling_ws(b, n, n_max, _lib.ptr(x), _lib.ptr(o), _lib.ptr(no), _lib.ptr(tmp), _lib.ptr(idx), _lib.ptr(ws),
                                             ctypes.c_size_t(need), st), "cbl_furthestsampling_ws")
    else:
        _lib.check(L.cbl_furthestsampling(b, n_max, _lib.ptr(x), _lib.ptr(o), _lib.ptr(no), _lib.ptr(tmp), _lib.ptr(idx), st), "cbl_furthestsampling")
    torch.cuda.synchronize()
    return idx.cpu().numpy(), tmp.cpu().numpy()


@pytest.mark.parametrize("n,b,seed", [(9000, 1, 2), (14000, 1, 3), (30000, 2, 4), (24000, 1, 6), (45000, 1, 5)])
def test_fps_dense_kernels_vs_oracle(n, b, seed):
    # the register / LDS / streaming dense variants stay reachable through the plain entry point
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(0, 3, (n, 3)).astype(np.float32)
    cuts = np.sort(rng.choice(np.arange(100, n - 100), b - 1, replace=False)) if b > 1 else np.array([], int)
    offset = np.concatenate([cuts, [n]]).astype(np.int32)
    noff = np.cumsum(np.minimum(np.diff(np.concatenate([[0], offset])) // 4, 400)).astype(np.int32)
    idx, tmp = _fps_raw(xyz, offset, noff, bucket=False)
    ridx, rtmp = O.furthestsampling(xyz, offset, noff)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(tmp, rtmp)


def _fps_chain(xyz, offset, strides, chain):
    """TransitionDown's sampling step applied stage after stage (pointops.fps_downsample) -> [(idx, new_offset, certificate)] per stage"""
    from contrastboundary_amd import pointops
    prev, pointops.fps_prefix_chain = pointops.fps_prefix_chain, chain
    try:
        p, o, out = dev(xyz), dev(np.int32(offset)), []
        for st in strides:
            new_p, new_o, idx = pointops.fps_downsample(p, o, st)
            out.append((idx.cpu().numpy(), new_o.cpu().numpy(), new_p._fps_certificate[0].cpu().numpy()))
            p, o = new_p, new_o
        return out
    finally:
        pointops.fps_prefix_chain = prev


def _oracle_chain(xyz, offset, strides):
    pts, off, out = xyz, np.int32(offset), []
    for st in strides:
        lens = np.diff(np.concatenate([[0], off]))
        noff = np.cumsum(lens // st).astype(np.int32)
        idx, _ = O.furthestsampling(pts, off, noff)
        out.append((idx, noff))
        pts, off = pts[idx], noff
    return out


def test_fps_chain_of_the_network_is_answered_by_prefixes():
    """the four sampling stages of the network (40960 -> 10240 -> 2560 -> 640 -> 160, blocks.py:61-68), two clouds: every stage bit for bit the oracle's FPS of
    that stage's input; on float data no arg-max is tied, so stage 1 certifies all of its samples and stages 2-4 are prefixes (their certificate is stage
    1's, handed on: the dense kernels behind stages 3 and 4 would have written 0)"""
    from contrastboundary_amd import synthetic as S
    xyz = np.concatenate([S.s_room(40960, seed=0)[0], S.s_room(20000, seed=1)[0]])
    offset, strides = [40960, 60960], [4, 4, 4, 4]
    want = _oracle_chain(xyz, offset, strides)
    got = _fps_chain(xyz, offset, strides, chain=True)
    plain = _fps_chain(xyz, offset, strides, chain=False)
    for (ridx, rnoff), (idx, noff, cert), (pidx, _, pcert) in zip(want, got, plain):
        np.testing.assert_array_equal(noff, rnoff)
        np.testing.assert_array_equal(idx, ridx)
        np.testing.assert_array_equal(pidx, ridx)
    m1 = np.diff(np.concatenate([[0], got[0][1]]))
    np.testing.assert_array_equal(got[0][2], (m1 + 1) // 2)                   # the tracked half of stage 1's picks were unique maxima (first real ties: picks 6135 / 4900)
    for stage in got[1:]:
        np.testing.assert_array_equal(stage[2], (m1 + 1) // 2)                # handed on: these stages did not run a sampler
    assert (plain[2][2] == 0).all() and (plain[3][2] == 0).all()              # without the chain the dense kernels run and certify nothing


def test_fps_chain_under_ties_runs_the_sampler():
    """a lattice ties every distance: the certificate ends at the first tied pick and the later stages sample for real, still the oracle's samples"""
    g = np.arange(24, dtype=np.float32)
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    rng = np.random.default_rng(0)
    xyz = np.concatenate([lat, rng.uniform(0, 3, (9000, 3)).astype(np.float32)])
    offset, strides = [len(lat), len(lat) + 9000], [3, 3, 3]
    want = _oracle_chain(xyz, offset, strides)
    got = _fps_chain(xyz, offset, strides, chain=True)
    for (ridx, rnoff), (idx, noff, cert) in zip(want, got):
        np.testing.assert_array_equal(idx, ridx)
    assert got[0][2][0] < 8                                                   # the lattice cloud ties within its first picks
    assert got[0][2][1] >= 1000                                               # the float cloud of the same batch is certified far enough for stage 2 ...
    assert got[1][2][1] == got[0][2][1]                                       # ... and handed on, while the lattice cloud beside it was sampled again


def test_fps_certificate_ends_exactly_at_the_first_tied_pick():
    """two coincident points tie when they become the maximum: the certificate is that pick's index; a request up to it is a prefix, one more runs the
    sampler — both the oracle's answer"""
    from contrastboundary_amd import pointops
    rng = np.random.default_rng(3)
    xyz = rng.uniform(0, 2, (20000, 3)).astype(np.float32)
    n, m1 = len(xyz), 6000
    pre, _ = O.furthestsampling(xyz, [n], [m1])
    spare = int(np.setdiff1d(np.arange(n), pre)[0])                           # a point the run never picks ...
    xyz[spare] = xyz[pre[1500]]                                               # ... becomes the twin of pick 1500: both reach the maximum together, there
    ridx1, _ = O.furthestsampling(xyz, [n], [m1])
    np.testing.assert_array_equal(ridx1[:1500], pre[:1500])
    hit = np.nonzero((ridx1 == pre[1500]) | (ridx1 == spare))[0]
    assert len(hit) and hit[0] == 1500                                        # inside the tracked half (3000)
    jstar = int(hit[0])
    idx1, cert1 = pointops._furthestsampling_raw(dev(xyz), dev(np.int32([n])), dev(np.int32([m1])), n, m1, want_cert=True)
    np.testing.assert_array_equal(idx1.cpu().numpy(), ridx1)
    assert cert1.cpu().tolist() == [jstar]
    p1 = dev(xyz[ridx1])
    for m2 in (jstar, jstar + 1):
        ridx2, _ = O.furthestsampling(xyz[ridx1], [m1], [m2])
        idx2, cert2 = pointops._furthestsampling_raw(p1, dev(np.int32([m1])), dev(np.int32([m2])), m1, m2, cert_in=cert1, want_cert=True)
        np.testing.assert_array_equal(idx2.cpu().numpy(), ridx2)
        if m2 == jstar:
            np.testing.assert_array_equal(ridx2, np.arange(jstar))               # the prefix property itself, on the oracle
            assert cert2.cpu().tolist() == [jstar]                             # handed on
        else:
            assert cert2.cpu().tolist()[0] <= jstar + 1                       # sampled for real (the pair ties again at the same pick)


def test_fps_bucket_full_size_room():
    # C2 / C4 shape: 40960 -> 10240 on the S-room scene; bucket-pruned kernel == dense kernel == oracle, running distances included
    from contrastboundary_amd import synthetic as S
    xyz, _ = S.s_room(40960, seed=0)
    idx_b, tmp_b = _fps_raw(xyz, [40960], [10240], bucket=True)
    idx_d, tmp_d = _fps_raw(xyz, [40960], [10240], bucket=False)
    np.testing.assert_array_equal(idx_b, idx_d)
    np.testing.assert_array_equal(tmp_b, tmp_d)
    ridx, rtmp = O.furthestsampling(xyz, [40960], [10240])
    np.testing.assert_array_equal(idx_b, ridx)
    np.testing.assert_array_equal(tmp_b, rtmp)
    assert len(np.unique(idx_b)) == 10240


def test_fps_bucket_ties_and_ragged_batches():
    # a 24^3 lattice (every distance tied, reference block size 1024 -> the bit-reversed thread rule decides) ...
    g = np.arange(24, dtype=np.float32)
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    idx, tmp = _fps_raw(lat, [len(lat)], [len(lat) // 3], bucket=True)
    ridx, rtmp = O.furthestsampling(lat, [len(lat)], [len(lat) // 3])
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(tmp, rtmp)
    # ... and a ragged batch: large, tiny (fewer points than a bucket), lattice, large; one cloud samples nothing new beyond its first point
    rng = np.random.default_rng(11)
    parts = [rng.uniform(0, 4, (15000, 3)), rng.uniform(0, 1, (37, 3)), lat[:9000], rng.normal(0, 1, (26000, 3))]
    xyz = np.concatenate(parts).astype(np.float32)
    offset = np.cumsum([len(p) for p in parts]).astype(np.int32)
    noff = np.cumsum([1500, 1, 3000, 2000]).astype(np.int32)
    idx, tmp = _fps_raw(xyz, offset, noff, bucket=True)
    ridx, rtmp = O.furthestsampling(xyz, offset, noff)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(tmp, rtmp)


def test_k3_k10_golden(P):
    g = lambda f: K310[f]
    inp = dev(g("grouping/input")).requires_grad_(True)
    out = P.grouping(inp, dev(g("grouping/idx")))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g("grouping/output"))
    out.backward(dev(g("grouping/grad_output")))
    np.testing.assert_allclose(inp.grad.cpu().numpy(), g("grouping/grad_input"), rtol=0, atol=ATOL)

    a = dev(g("subtraction/input1")).requires_grad_(True); b = dev(g("subtraction/input2")).requires_grad_(True)
    out = P.subtraction(a, b, dev(g("subtraction/idx")))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g("subtraction/output"))
    out.backward(dev(g("subtraction/grad_output")))
    np.testing.assert_array_equal(a.grad.cpu().numpy(), g("subtraction/grad_input1"))     # in-register sum, fixed order
    np.testing.assert_allclose(b.grad.cpu().numpy(), g("subtraction/grad_input2"), rtol=0, atol=ATOL)

    x = dev(g("aggregation/input")).requires_grad_(True); pos = dev(g("aggregation/position")).requires_grad_(True)
    w = dev(g("aggregation/weight")).requires_grad_(True)
    out = P.aggregation(x, pos, w, dev(g("aggregation/idx")))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g("aggregation/output"))
    out.backward(dev(g("aggregation/grad_output")))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g("aggregation/grad_input"), rtol=0, atol=ATOL)
    np.testing.assert_array_equal(pos.grad.cpu().numpy(), g("aggregation/grad_position"))
    np.testing.assert_allclose(w.grad.cpu().numpy(), g("aggregation/grad_weight"), rtol=0, atol=ATOL)


def _call_interp(P, inp, idx, w, go):
    """K5/K6 through the raw C ABI (the python-level Interpolation recomputes idx/weight itself)"""
    import ctypes
    from contrastboundary_amd import _lib
    n, k = idx.shape; c = inp.shape[1]
    out = torch.zeros((n, c), device="cuda")
    L = _lib.lib()
    _lib.check(L.cbl_interpolation_forward(n, c, k, _lib.ptr(inp), _lib.ptr(idx), _lib.ptr(w), _lib.ptr(out), _lib.stream_of(inp)), "fwd")
    gi = torch.zeros_like(inp)
    _lib.check(L.cbl_interpolation_backward(n, c, k, _lib.ptr(go), _lib.ptr(idx), _lib.ptr(w), _lib.ptr(gi), _lib.stream_of(inp)), "bwd")
    return out, gi


def test_interpolation_kernels_golden(P):
    g = lambda f: K310[f"interpolation/{f}"]
    out, gi = _call_interp(P, dev(g("input")), dev(g("idx")), dev(g("weight")), dev(g("grad_output")))
    np.testing.assert_array_equal(out.cpu().numpy(), g("output"))
    np.testing.assert_allclose(gi.cpu().numpy(), g("grad_input"), rtol=0, atol=ATOL)


@pytest.mark.parametrize("c", [3, 13, 32, 64])
def test_gather_family_vs_oracle(P, c):
    rng = np.random.default_rng(c)
    n, m, ns = 1000, 777, 16
    wc = 4 if c % 4 == 0 else 1
    inp = rng.normal(size=(n, c)).astype(np.float32); idx = rng.integers(0, n, (m, ns)).astype(np.int32)
    np.testing.assert_array_equal(P.grouping(dev(inp), dev(idx)).cpu().numpy(), O.grouping_forward(inp, idx))
    go = rng.normal(size=(m, ns, c)).astype(np.float32)
    t = dev(inp).requires_grad_(True); P.grouping(t, dev(idx)).backward(dev(go))
    np.testing.assert_allclose(t.grad.cpu().numpy(), O.grouping_backward(go, idx, n), rtol=0, atol=ATOL)
    # subtraction / aggregation use idx (n, ns)
    idx2 = rng.integers(0, n, (n, ns)).astype(np.int32); b2 = rng.normal(size=(n, c)).astype(np.float32)
    np.testing.assert_array_equal(P.subtraction(dev(inp), dev(b2), dev(idx2)).cpu().numpy(), O.subtraction_forward(inp, b2, idx2))
    pos = rng.normal(size=(n, ns, c)).astype(np.float32); w = rng.normal(size=(n, ns, wc)).astype(np.float32)
    np.testing.assert_array_equal(P.aggregation(dev(inp), dev(pos), dev(w), dev(idx2)).cpu().numpy(), O.aggregation_forward(inp, pos, w, idx2))
    # K5/K6
    k = 3; iidx = rng.integers(0, n, (m, k)).astype(np.int32); iw = rng.uniform(size=(m, k)).astype(np.float32)
    igo = rng.normal(size=(m, c)).astype(np.float32)
    out, gi = _call_interp(P, dev(inp), dev(iidx), dev(iw), dev(igo))
    np.testing.assert_array_equal(out.cpu().numpy(), O.interpolation_forward(inp, iidx, iw))
    np.testing.assert_allclose(gi.cpu().numpy(), O.interpolation_backward(igo, iidx, iw, n), rtol=0, atol=ATOL)


def test_queryandgroup_and_interpolation_composites(P):
    rng = np.random.default_rng(7)
    n, m, c, k = 2000, 500, 32, 16
    xyz = rng.uniform(size=(n, 3)).astype(np.float32); feat = rng.normal(size=(n, c)).astype(np.float32)
    off = np.int32([1200, 2000]); sel = np.concatenate([np.arange(0, 1200, 4), np.arange(1200, 2000, 4)]); q = xyz[sel]; noff = np.int32([300, 500])
    out = P.queryandgroup(k, dev(xyz), dev(q), dev(feat), None, dev(off), dev(noff), use_xyz=True)
    ridx, _ = O.knnquery(k, xyz, q, off, noff)
    ref = np.concatenate([xyz[ridx] - q[:, None, :], feat[ridx]], -1)          # pointops.py:90-98
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    out2 = P.queryandgroup(k, dev(xyz), dev(q), dev(feat), dev(ridx), dev(off), dev(noff), use_xyz=False)
    np.testing.assert_array_equal(out2.cpu().numpy(), feat[ridx])
    # interpolation: coarse (q, fq) -> fine xyz, k=3 and k=1                    pointops.py:164-178
    fq = rng.normal(size=(m, c)).astype(np.float32)
    for kk in (3, 1):
        got = P.interpolation(dev(q), dev(xyz), dev(fq), dev(noff), dev(off), k=kk).cpu().numpy()
        iidx, id2 = O.knnquery(kk, q, xyz, noff, off)
        dist = np.sqrt(id2); rec = (np.float32(1.0) / (dist + np.float32(1e-8))).astype(np.float32)
        norm = np.zeros((n, 1), np.float32)
        for i in range(kk):
            norm[:, 0] += rec[:, i]
        w = (rec / norm).astype(np.float32)
        np.testing.assert_array_equal(got, O.interpolation_forward(fq, iidx, w))


@pytest.mark.parametrize("m,k,c", [(500, 16, 64), (37, 5, 4), (129, 7, 12), (1, 3, 8), (64, 16, 128), (33, 9, 132), (50, 4, 6)])
def test_queryandgroup_shapes(P, m, k, c):
    """F1 with the index given: the aligned-piece kernel (16 output rows through LDS), its element-wise tail, and the fall-backs
    (channel counts that are not a multiple of 4, wide rows)"""
    rng = np.random.default_rng(m * 1000 + k * 10 + c)
    n = 300
    xyz = rng.uniform(size=(n, 3)).astype(np.float32); feat = rng.normal(size=(n, c)).astype(np.float32)
    q = rng.uniform(size=(m, 3)).astype(np.float32)
    idx = rng.integers(0, n, (m, k)).astype(np.int32)
    off = np.int32([n]); noff = np.int32([m])
    for use_xyz in (True, False):
        out = P.queryandgroup(k, dev(xyz), dev(q), dev(feat), dev(idx), dev(off), dev(noff), use_xyz=use_xyz).cpu().numpy()
        ref = np.concatenate([xyz[idx] - q[:, None, :], feat[idx]], -1) if use_xyz else feat[idx]      # pointops.py:90-98
        np.testing.assert_array_equal(out, ref)


def test_full_size_properties(P):
    """BASELINE C2 size (N=40960, K=16, C=64): size-independent properties instead of the O(N^2) oracle."""
    rng = np.random.default_rng(0)
    n, k, c = 40960, 16, 64
    xyz = dev(rng.uniform(0, 2.05, (n, 3)).astype(np.float32)); o = dev(np.int32([n]))
    idx, d2 = P.knnquery_raw(k, xyz, xyz, o, o)
    assert (d2[:, 1:] >= d2[:, :-1]).all()                       # ascending
    assert (idx[:, 0] == torch.arange(n, device="cuda")).all() and (d2[:, 0] == 0).all()   # self is nearest
    assert int(idx.min()) >= 0 and int(idx.max()) < n
    # recomputing d2 from idx reproduces the stored distances bit for bit
    nb = xyz[idx.long()]; dd = xyz[:, None, :] - nb
    re = (dd[..., 0] * dd[..., 0] + dd[..., 1] * dd[..., 1]) + dd[..., 2] * dd[..., 2]
    assert torch.equal(re, d2)
    # exact and auto paths agree on the whole scene
    idx_e, d2_e = P.knnquery_raw(k, xyz, xyz, o, o, algo="exact")
    assert torch.equal(idx, idx_e) and torch.equal(d2, d2_e)
    # a random 256-query sample against the oracle
    sel = np.sort(rng.choice(n, 256, replace=False))
    ridx, _ = O.knnquery(k, xyz.cpu().numpy(), xyz.cpu().numpy()[sel], [n], [256])
    np.testing.assert_array_equal(idx.cpu().numpy()[sel], ridx)
    # grouping: gather == torch indexing; backward of ones == neighbour in-degree (linearity / checksum)
    feat = dev(rng.normal(size=(n, c)).astype(np.float32)).requires_grad_(True)
    grouped = P.grouping(feat, idx)
    assert torch.equal(grouped, feat[idx.long()])
    grouped.backward(torch.ones_like(grouped))
    deg = torch.bincount(idx.view(-1).long(), minlength=n).float()
    assert torch.equal(feat.grad, deg[:, None].expand(-1, c))


def test_knn_set_variant(P):
    """cbl_knnquery_set: identical distances and neighbour sets; only the order inside groups of equal distance is free"""
    g = np.arange(9, dtype=np.float32)
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    rng = np.random.default_rng(0)
    rnd = rng.uniform(0, 2, (5000, 3)).astype(np.float32)
    rnd[100:110] = rnd[50:60]                                   # duplicated points -> ties
    for xyz, k in ((np.concatenate([lat] * 4), 7), (rnd, 16), (rnd, 36)):
        n = len(xyz); o = dev(np.int32([n]))
        idx, d2 = P.knnquery_raw(k, dev(xyz), dev(xyz), o, o, algo="set")
        ridx, rd2 = O.knnquery(k, xyz, xyz, [n], [n])
        np.testing.assert_array_equal(d2.cpu().numpy().view(np.uint32), rd2.view(np.uint32))
        np.testing.assert_array_equal(np.sort(idx.cpu().numpy(), 1), np.sort(ridx, 1))


def test_empty_and_degenerate_inputs(P):
    """no queries, an empty cloud inside a batch, one-point clouds: nothing to compute is not an error (and not a crash)"""
    rng = np.random.default_rng(0)
    xyz = dev(rng.uniform(0, 1, (300, 3)).astype(np.float32))
    o = dev(np.int32([100, 100, 300]))                                 # cloud 1 is empty
    q = dev(rng.uniform(0, 1, (50, 3)).astype(np.float32))
    qo = dev(np.int32([20, 20, 50]))
    idx, d2 = P.knnquery_raw(4, xyz, q, o, qo)
    ridx, rd2 = O.knnquery(4, xyz.cpu().numpy(), q.cpu().numpy(), [100, 100, 300], [20, 20, 50])
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    idx0, d0 = P.knnquery_raw(4, xyz, q[:0].contiguous(), o, dev(np.int32([0, 0, 0])))
    assert idx0.shape == (0, 4) and d0.shape == (0, 4)
    feat = dev(rng.normal(size=(300, 8)).astype(np.float32))
    assert P.grouping(feat, idx0).shape == (0, 4, 8)
    fidx = P.furthestsampling(xyz, o, dev(np.int32([10, 10, 40])))     # the empty cloud contributes no sample
    ridx, _ = O.furthestsampling(xyz.cpu().numpy(), [100, 100, 300], [10, 10, 40])
    np.testing.assert_array_equal(fidx.cpu().numpy(), ridx)
    one = dev(np.float32([[1, 2, 3]]))
    i1, d1 = P.knnquery_raw(3, one, one, dev(np.int32([1])), dev(np.int32([1])))
    r1, _ = O.knnquery(3, one.cpu().numpy(), one.cpu().numpy(), [1], [1])
    np.testing.assert_array_equal(i1.cpu().numpy(), r1)


@pytest.mark.parametrize("algo", ["auto", "set"])
def test_knn_large_k_with_ties_falls_back_to_the_reference_order(P, algo):
    # 64 < K: block-select kernel; on a lattice every query has ties at the K-th distance, so every query is redone by the exact kernel
    g = np.arange(17, dtype=np.float32)
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)       # 4913 points
    rng = np.random.default_rng(3)
    pts = np.concatenate([lat, rng.uniform(0, 16, (3000, 3)).astype(np.float32)])
    q = pts[rng.choice(len(pts), 200, replace=False)]
    o, qo = np.int32([len(pts)]), np.int32([200])
    idx, d2 = P.knnquery_raw(100, dev(pts), dev(q), dev(o), dev(qo), algo=algo)
    ridx, rd2 = O.knnquery(100, pts, q, o, qo)
    np.testing.assert_array_equal(d2.cpu().numpy().view(np.uint32), rd2.view(np.uint32))
    if algo == "auto":
        np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    else:   # same neighbour set per query
        np.testing.assert_array_equal(np.sort(idx.cpu().numpy(), 1), np.sort(ridx, 1))


@pytest.mark.parametrize("K", [16, 36, 100])
def test_knn_anytie_policy_never_replays_and_keeps_the_distances(P, K):
    # lattice + duplicates: every query has ties.  'anytie' must give the reference's distances (bit for bit) and valid neighbours,
    # and 'reference' must stay bit-exact on the same data
    g = np.arange(17, dtype=np.float32)
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    pts = np.concatenate([lat, lat[:500]])                                # + exact duplicates
    o = np.int32([len(pts)])
    idx, d2 = P.knnquery_raw(K, dev(pts), dev(pts), dev(o), dev(o), algo="anytie")
    ridx, rd2 = O.knnquery(K, pts, pts, o, o)
    np.testing.assert_array_equal(d2.cpu().numpy().view(np.uint32), rd2.view(np.uint32))
    ii = idx.cpu().numpy().astype(np.int64)
    dd = ((pts[ii] - pts[:, None, :]) ** 2)
    np.testing.assert_array_equal(((dd[..., 0] + dd[..., 1]) + dd[..., 2]).astype(np.float32).view(np.uint32), rd2.view(np.uint32))   # idx really at those distances
    assert all(len(set(r)) == K for r in ii[::97])                        # no neighbour twice
    # the rule of its own: the K smallest by (distance, index)
    allq = np.arange(0, len(pts), 53)
    dq = ((pts[None, :, :] - pts[allq][:, None, :]) ** 2)
    dq = ((dq[..., 0] + dq[..., 1]) + dq[..., 2]).astype(np.float32)
    canon = np.lexsort((np.broadcast_to(np.arange(len(pts)), dq.shape), dq), axis=1)[:, :K]
    np.testing.assert_array_equal(ii[allq], canon)
    prev = P.set_knn_tie_policy("anytie")
    try:
        idx2, _ = P.knnquery_raw(K, dev(pts), dev(pts), dev(o), dev(o))   # 'auto' follows the policy
        assert torch.equal(idx2, idx)
    finally:
        P.set_knn_tie_policy(prev)
    idx3, _ = P.knnquery_raw(K, dev(pts), dev(pts), dev(o), dev(o))
    np.testing.assert_array_equal(idx3.cpu().numpy(), ridx)


@pytest.mark.parametrize("seed", range(6))
def test_fps_speculative_rounds_vs_dense_kernels(seed):
    """the bucket kernel's speculative rounds (several samples per barrier, validated prefix by prefix) against the dense kernels — an
    independent implementation of the same arg-max sequence — on ragged batches of rooms and uniform clouds: indices and final tmp bit-exact"""
    from contrastboundary_amd import synthetic as S
    rng = np.random.default_rng(100 + seed)
    b = int(rng.integers(1, 5))
    sizes = [int(rng.integers(3072, 30000)) for _ in range(b)]
    clouds = []
    for i, n in enumerate(sizes):
        clouds.append(S.s_room(n, seed=10 * seed + i)[0] if (seed + i) % 2 == 0 else S.s_uniform(n, seed=10 * seed + i))
    xyz = np.concatenate(clouds).astype(np.float32)
    offset = np.cumsum(sizes).astype(np.int32)
    noff = np.cumsum([max(1, int(n / rng.choice([2, 4, 4, 7]))) for n in sizes]).astype(np.int32)
    idx_b, tmp_b = _fps_raw(xyz, offset, noff, bucket=True)
    idx_d, tmp_d = _fps_raw(xyz, offset, noff, bucket=False)
    np.testing.assert_array_equal(idx_b, idx_d)
    np.testing.assert_array_equal(tmp_b.view(np.uint32), tmp_d.view(np.uint32))


def test_block_candidates_of_a_grid_search_cover_its_neighbours():
    """cbl_knn_grid_block_candidates (bench.py roofline.search: "pairs visited"): per query the number of supports in its 27-cell block, recomputed from the
    grid the search left in the workspace — at least the query itself, never more than its cloud, for nearly every query at least K (the block certifies
    the list) and on average a few cells' worth of points, not the cloud"""
    import torch
    from contrastboundary_amd import pointops, synthetic as S
    n, K = 20000, 36
    xyz = torch.from_numpy(S.s_room(n, seed=4)[0]).cuda(); o = torch.tensor([n // 2, n], dtype=torch.int32, device="cuda")
    cnt = pointops.knn_block_candidates(K, xyz, o, "set")
    torch.cuda.synchronize()
    assert cnt.shape == (n,) and int(cnt.min()) >= 1 and int(cnt.max()) <= n // 2
    assert float((cnt >= K).float().mean()) > 0.98
    assert K <= float(cnt.float().mean()) <= 40 * K                       # a few cells' worth of points per query, not the cloud (n / 2 = 10000)
