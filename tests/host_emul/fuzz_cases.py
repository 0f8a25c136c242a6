"""TEST INFRASTRUCTURE: randomized cases for the entry points of the whole-library host build (tests/host_emul/full_library.py) against the oracles — shapes on both
sides of every dispatch threshold, ragged batches with one-point clouds, clouds smaller than K, coincident points, lattices and planes (exactly tied distances),
shadow entries, hub targets, ignored labels, every contrast flavour.  tests/test_fuzz_host.py runs a bounded, seeded share of them; a campaign is

    python tests/host_emul/fuzz_cases.py <knn|radius|grid|subsample|pyramid|fps|transpose|cbl|gather|aggregation|attention> <seed> <cases>

What the round-5 campaign (a few thousand cases) found is in DESIGN.md 5: two numerical defects of the contrast kernels in extreme regimes (fixed, regression tests in
tests/test_cbl_host.py) and one limit of the emulation (K > 64 of the brute-force search keeps its heap in LDS and every lane of the wave replays the same update —
lockstep on the device, sequential here; that path is held on the device by tests/test_gpu_pointops.py)."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cbl_oracle as C                                    # noqa: E402
from tests import oracle_lib as O                                     # noqa: E402
from tests.host_emul import full_library                              # noqa: E402

F = ctypes.c_float
_L = None


def lib():
    global _L
    if _L is None:
        _L = full_library.load()
        for name in ("cbl_knnquery_workspace_bytes", "cbl_grid_subsampling_workspace_bytes", "cbl_radius_neighbors_workspace_bytes", "cbl_furthestsampling_workspace_bytes",
                     "cbl_neighbor_transpose_workspace_bytes"):
            getattr(_L, name).restype = ctypes.c_size_t
    return _L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def aligned(shape, dtype=np.float32):
    n = int(np.prod(shape))
    raw = np.zeros(n * np.dtype(dtype).itemsize + 16, np.uint8)
    off = (-raw.ctypes.data) % 16
    return raw[off:off + n * np.dtype(dtype).itemsize].view(dtype).reshape(shape)


def contract(idx, n, order_dst=None):
    """the transposed table by its definition (include/cbl_amd.h: cbl_neighbor_transpose)"""
    flat = idx.reshape(-1)
    keep = np.nonzero((flat >= 0) & (flat < n))[0]
    tgt = flat[keep]
    if order_dst is not None:
        pos = np.empty(n, np.int64); pos[order_dst] = np.arange(n)
        tgt = pos[tgt]
    o = np.argsort(tgt, kind="stable")
    inv_start = np.zeros(n + 1, np.int64)
    np.add.at(inv_start, tgt + 1, 1)
    return np.cumsum(inv_start).astype(np.int32), keep[o].astype(np.int32)

def cloud(rng, n, kind):
    if kind == 0: return rng.uniform(0, 1, (n, 3)).astype(np.float32)
    if kind == 1: return (rng.integers(0, 6, (n, 3)) * 0.125).astype(np.float32)      # heavy ties + duplicates
    if kind == 2: return np.concatenate([rng.uniform(0, 1, (n, 2)), np.zeros((n, 1))], 1).astype(np.float32)   # plane
    if kind == 3: return np.repeat(rng.uniform(0, 1, (1, 3)), n, 0).astype(np.float32)   # all coincident
    return (rng.normal(size=(n, 3)) * np.float32([1, 0.01, 100])).astype(np.float32)   # anisotropic

def knn_case(rng, it):
    b = int(rng.integers(1, 4))
    big = rng.random() < 0.5
    sizes = [int(rng.integers(1, 40)) for _ in range(b)]
    if big: sizes[int(rng.integers(0, b))] = int(rng.integers(2048, 2600))
    K = int(rng.choice([1, 2, 3, 8, 16, 17, 32, 36, 63, 64]))
    kind = int(rng.integers(0, 5))
    xyz = np.concatenate([cloud(rng, n, kind) + 3.0 * i for i, n in enumerate(sizes)])
    off = np.cumsum(sizes).astype(np.int32)
    selfq = rng.random() < 0.6
    if selfq:
        q, qoff = xyz, off
    else:
        qs = [int(rng.integers(0, 30)) for _ in range(b)]
        if sum(qs) == 0: qs[0] = 1
        q = np.concatenate([cloud(rng, m, kind) + 3.0 * i for i, m in enumerate(qs)]).astype(np.float32).reshape(-1, 3)
        qoff = np.cumsum(qs).astype(np.int32)
    n, m = len(xyz), len(q)
    ridx, rd2 = O.knnquery(K, xyz, q, off, qoff)
    idx, d2 = np.full((m, K), -7, np.int32), np.full((m, K), np.nan, np.float32)
    nbytes = lib().cbl_knnquery_workspace_bytes(b, n, m, K)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = lib().cbl_knnquery(b, n, m, K, P(xyz), P(q), P(off), P(qoff), P(idx), P(d2), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0, ("knn rc", rc, sizes, K, kind, selfq)
    if not (np.array_equal(idx, ridx) and np.array_equal(d2.view(np.uint32), rd2.view(np.uint32))):
        bad = np.nonzero((idx != ridx).any(1) | (d2.view(np.uint32) != rd2.view(np.uint32)).any(1))[0]
        print("KNN MISMATCH", it, sizes, K, kind, selfq, "rows", bad[:5], idx[bad[0]], ridx[bad[0]], d2[bad[0]], rd2[bad[0]]); return False
    return True

def radius_case(rng, it):
    b = int(rng.integers(1, 4))
    sizes = [int(rng.integers(1, 400)) for _ in range(b)]
    qs = [int(rng.integers(0, 100)) for _ in range(b)]
    if sum(qs) == 0: qs[0] = 1
    kind = int(rng.integers(0, 5))
    s = np.concatenate([cloud(rng, n, kind) for n in sizes]); q = np.concatenate([cloud(rng, m, kind) for m in qs]).reshape(-1, 3).astype(np.float32)
    if rng.random() < 0.5: q, qs = s, sizes
    sl, ql = np.int32(sizes), np.int32(qs)
    r = float(rng.choice([0.05, 0.2, 0.5, 3.0])); limit = int(rng.choice([1, 2, 7, 16, 33, 64]))
    nq, ns = len(q), len(s)
    qo, so = np.cumsum(ql).astype(np.int32), np.cumsum(sl).astype(np.int32)
    out, counts, mx = np.full((nq, limit), -1, np.int32), np.full(nq, -1, np.int32), np.zeros(1, np.int32)
    nbytes = lib().cbl_radius_neighbors_workspace_bytes(b, ns)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = lib().cbl_radius_neighbors(b, nq, ns, P(q), P(s), P(qo), P(so), F(r), limit, P(out), P(counts), P(mx), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0, ("radius rc", rc)
    ref, rcounts, mc = O.radius_neighbors(q, s, ql, sl, r, limit)
    if not (np.array_equal(out, ref) and np.array_equal(counts, rcounts) and int(mx[0]) == mc):
        bad = np.nonzero((out != ref).any(1) | (counts != rcounts))[0]
        print("RADIUS MISMATCH", it, sizes, qs, kind, r, limit, "rows", bad[:5], out[bad[0]] if len(bad) else None, ref[bad[0]] if len(bad) else None, int(mx[0]), mc); return False
    return True

def grid_case(rng, it):
    b = int(rng.integers(1, 4))
    sizes = [int(rng.integers(1, 600)) for _ in range(b)]
    kind = int(rng.integers(0, 5))
    xyz = np.concatenate([cloud(rng, n, kind) for n in sizes])
    lens = np.int32(sizes); off = np.cumsum(lens).astype(np.int32)
    dl = float(rng.choice([0.01, 0.1, 0.3, 5.0]))
    n = len(xyz)
    op = np.full((n, 3), np.nan, np.float32); out_len, total = np.full(b, -1, np.int32), np.full(1, -1, np.int32)
    nbytes = lib().cbl_grid_subsampling_workspace_bytes(b, n)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = lib().cbl_grid_subsampling(b, n, P(xyz), P(off), F(dl), 0, None, 0, None, P(op), None, None, P(out_len), P(total), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0, ("grid rc", rc)
    rp, rl = O.grid_subsampling(xyz, lens, dl)
    m = int(total[0])
    if not (np.array_equal(out_len, rl) and m == len(rp) and np.array_equal(op[:m].view(np.uint32), rp.view(np.uint32))):
        print("GRID MISMATCH", it, sizes, kind, dl, m, len(rp), out_len, rl); return False
    return True

def subsample_case(rng, it):
    """grid subsampling with features and labels (barycentres of points and features, majority label per column — the deterministic rule of oracle/tfops_oracle.c
    where votes tie), cloud by cloud against the wrapper flavour of the oracle, bit for bit"""
    b = int(rng.integers(1, 4)); sizes = [int(rng.integers(1, 500)) for _ in range(b)]; kind = int(rng.integers(0, 5))
    xyz = np.concatenate([cloud(rng, n, kind) for n in sizes]); lens = np.int32(sizes); off = np.cumsum(lens).astype(np.int32); n = len(xyz)
    dl = float(rng.choice([0.01, 0.1, 0.3, 5.0])); fd = int(rng.choice([1, 3, 5])); ld = int(rng.choice([1, 2]))
    feat = rng.normal(size=(n, fd)).astype(np.float32); labels = rng.integers(0, int(rng.choice([2, 13])), (n, ld)).astype(np.int32)
    op, of, ol = np.full((n, 3), np.nan, np.float32), np.full((n, fd), np.nan, np.float32), np.full((n, ld), -1, np.int32)
    out_len, total = np.full(b, -1, np.int32), np.full(1, -1, np.int32)
    nbytes = lib().cbl_grid_subsampling_workspace_bytes(b, n); ws = np.zeros(nbytes + 64, np.uint8)
    rc = lib().cbl_grid_subsampling(b, n, P(xyz), P(off), F(dl), fd, P(feat), ld, P(labels), P(op), P(of), P(ol), P(out_len), P(total), P(ws), ctypes.c_size_t(nbytes), None)
    ok = rc == 0
    s = t = 0
    for c in range(b):
        e = s + int(lens[c])
        fp, ff, fl, _ = O.grid_subsampling_full(xyz[s:e], feat[s:e], labels[s:e], dl)
        k = fp.shape[0]
        ok = (ok and int(out_len[c]) == k and np.array_equal(op[t:t + k].view(np.uint32), fp.view(np.uint32)) and np.array_equal(of[t:t + k].view(np.uint32), ff.view(np.uint32))
              and np.array_equal(ol[t:t + k], fl))
        s, t = e, t + k
    if not ok:
        print("SUBSAMPLE MISMATCH", it, sizes, kind, dl, fd, ld, rc)
    return ok


def pyramid_case(rng, it):
    """the ConvNet's input pyramid as one native call (cbl_pyramid, /root/reference/tensorflow/datasets/base.py:767-842) on drawn scenes — 1 .. 3 clouds, 1 .. 4 layers,
    neighbourhood limits 1 .. 40 — against the oracle's operators applied layer by layer, every table bit for bit"""
    L = _full(["cbl_pyramid_layer_workspace_bytes"])

    def ptrs(arrs, count):
        a = (ctypes.c_void_p * count)()
        for i, x in enumerate(arrs):
            a[i] = x.ctypes.data
        return a
    b = int(rng.integers(1, 4)); sizes = [int(rng.integers(1, 900)) for _ in range(b)]; kind = int(rng.choice([0, 0, 1, 2, 4]))
    xyz = np.concatenate([cloud(rng, n, kind) for n in sizes]).astype(np.float32); lens = np.int32(sizes); n = len(xyz)
    layers = int(rng.integers(1, 5)); r0 = float(rng.choice([0.05, 0.12, 0.3])); dl0 = float(rng.choice([0.02, 0.05, 0.15]))
    limits = rng.integers(1, 41, layers).astype(np.int32)
    gb = L.cbl_radius_neighbors_workspace_bytes(b, n)
    grids = [np.zeros(gb + 64, np.uint8) for _ in range(layers)]
    nb = [np.full((n, int(limits[l])), -5, np.int32) for l in range(layers)]
    m1 = max(layers - 1, 1)
    pp = [np.full((n, 3), np.nan, np.float32) for _ in range(m1)]; pl = [np.full(b, -1, np.int32) for _ in range(m1)]
    po = [np.full((n, int(limits[min(l, layers - 1)])), -5, np.int32) for l in range(m1)]; up = [np.full((n, int(limits[min(l, layers - 1)])), -5, np.int32) for l in range(m1)]
    mx, sizes_out = np.full(3 * layers, -1, np.int32), np.full(layers, -1, np.int32)
    nbytes = L.cbl_pyramid_layer_workspace_bytes(b, n); ws = np.zeros(nbytes + 64, np.uint8)
    many = layers > 1
    rc = L.cbl_pyramid(b, n, P(xyz), P(lens), F(r0), F(dl0), layers, P(limits), ptrs(grids, layers), ctypes.c_size_t(gb), ptrs(nb, layers),
                       ptrs(pp, m1) if many else None, ptrs(pl, m1) if many else None, ptrs(po, m1) if many else None, ptrs(up, m1) if many else None,
                       P(mx), P(sizes_out), P(ws), ctypes.c_size_t(nbytes), None)
    ok = rc == 0
    pts, ln, r, dl = xyz, lens, r0, dl0
    for l in range(layers):
        if not ok:
            break
        m = pts.shape[0]; lim = int(limits[l])
        ref, _, mc = O.radius_neighbors(pts, pts, ln, ln, r, lim)
        ok = int(sizes_out[l]) == m and np.array_equal(nb[l][:m], ref) and int(mx[3 * l]) == mc
        if l == layers - 1 or not ok:
            break
        sub, sl = O.grid_subsampling(pts, ln, 2 * dl); k = sub.shape[0]
        ok = np.array_equal(pl[l], sl) and np.array_equal(pp[l][:k].view(np.uint32), sub.view(np.uint32))
        ref, _, mc = O.radius_neighbors(sub, pts, sl, ln, r, lim); ok = ok and np.array_equal(po[l][:k], ref) and int(mx[3 * l + 1]) == mc
        ref, _, mc = O.radius_neighbors(pts, sub, ln, sl, 2 * r, lim); ok = ok and np.array_equal(up[l][:m], ref) and int(mx[3 * l + 2]) == mc
        pts, ln, r, dl = np.ascontiguousarray(sub), sl.astype(np.int32), 2 * r, 2 * dl
    if not ok:
        print("PYRAMID MISMATCH", it, sizes, kind, layers, r0, dl0, limits, rc)
    return ok


def fps_case(rng, it):
    b = int(rng.integers(1, 4))
    sizes = [int(rng.integers(1, 700)) for _ in range(b)]
    if rng.random() < 0.25: sizes[0] = int(rng.integers(3072, 3600))
    kind = int(rng.integers(0, 5))
    xyz = np.concatenate([cloud(rng, n, kind) + 2.0 * i for i, n in enumerate(sizes)])
    ms = [int(rng.integers(1, n + 1)) if rng.random() < 0.5 else max(1, n // 4) for n in sizes]
    off, noff = np.cumsum(sizes).astype(np.int32), np.cumsum(ms).astype(np.int32)
    n = len(xyz); n_max = max(sizes)
    tmp, idx = np.full(n, 1e10, np.float32), np.full(int(noff[-1]), -1, np.int32)
    nbytes = lib().cbl_furthestsampling_workspace_bytes(b, n, n_max)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = lib().cbl_furthestsampling_ws(b, n, n_max, P(xyz), P(off), P(noff), P(tmp), P(idx), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0, ("fps rc", rc)
    ref, _ = O.furthestsampling(xyz, off, noff, n_max)
    if not np.array_equal(idx, ref):
        bad = np.nonzero(idx != ref)[0]
        print("FPS MISMATCH", it, sizes, ms, kind, "first", bad[:5], idx[bad[:5]], ref[bad[:5]]); return False
    return True


def transpose_case(rng, it):
    """cbl_neighbor_transpose against its contract, then K4 as a gather over the table against np.add.at in the reference loop's order (bit for bit)"""
    n = int(rng.choice([1, 2, 63, 64, 65, 300, 1000, 2500]))
    m = n if rng.random() < 0.5 else int(rng.choice([1, 5, 64, 200, 1500]))
    K = int(rng.choice([1, 2, 3, 8, 16, 36, 64]))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        idx = rng.integers(0, n, (m, K))
    elif kind == 1:
        idx = rng.integers(0, n + 1, (m, K))                          # shadow entries
    elif kind == 2:
        idx = np.full((m, K), int(rng.integers(0, n)))                # one hub
    else:
        idx = np.minimum(rng.geometric(0.05, (m, K)) - 1, n)          # skewed towards low targets, with shadows
    idx = np.ascontiguousarray(idx, np.int32)
    use_order = rng.random() < 0.5
    od = rng.permutation(n).astype(np.int32) if use_order else None
    osrc = rng.permutation(m).astype(np.int32) if (use_order and rng.random() < 0.5) else None
    inv_start, inv_src = np.full(n + 1, -1, np.int32), np.full(m * K, -1, np.int32)
    nbytes = lib().cbl_neighbor_transpose_workspace_bytes(m, n, K)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = lib().cbl_neighbor_transpose(m, n, K, P(idx), P(osrc), P(od), P(inv_start), P(inv_src), P(ws), ctypes.c_size_t(nbytes), None)
    rs, rsrc = contract(idx, n, od)
    if not (rc == 0 and np.array_equal(inv_start, rs) and np.array_equal(inv_src[:rs[-1]], rsrc)):
        print("TABLE MISMATCH", it, n, m, K, kind, use_order, osrc is not None, rc)
        return False
    c = int(rng.choice([1, 3, 4, 32, 33]))
    go = rng.normal(size=(m, K, c)).astype(np.float32)
    gi = np.full((n, c), np.nan, np.float32)
    rc = lib().cbl_grouping_backward_csr(n, c, P(go), P(od), P(inv_start), P(inv_src), P(gi), None)
    ref = np.zeros((n, c), np.float32)
    flat = idx.reshape(-1); keep = (flat >= 0) & (flat < n)
    np.add.at(ref, flat[keep], go.reshape(-1, c)[keep])
    if not (rc == 0 and np.array_equal(gi, ref)):
        print("K4 MISMATCH", it, n, m, K, kind, c, use_order, rc)
        return False
    return True


def cbl_case(rng, it, extreme=True):
    """the pair-mining kernels (cbl_contrast_pairs_forward + _backward over a numpy-built transposed table) against oracle/cbl_oracle.py: pytorch and TF heads,
    'softnn' / 'nce', margin 'S', ignored labels, shadow neighbours, coincident features.  extreme: feature scales and temperatures at which the exponentials
    span 30 orders of magnitude — the default flavour ('softnn') must hold there too; the 'nce' / 'S' flavours are drawn in the moderate regime only (beyond it
    oracle and kernel both produce NaN or denormal-dependent values)."""
    n = int(rng.choice([2, 3, 17, 64, 65, 200, 700]))
    nsample = int(rng.choice([2, 3, 5, 9, 17, 18, 33, 34, 64, 65]))
    d = int(rng.choice([4, 8, 16, 32, 64]))
    ncls = int(rng.choice([2, 3, 13]))
    tf = rng.random() < 0.5
    nce = rng.random() < 0.3
    sep = tf and rng.random() < 0.3
    moderate = (nce or sep) or not extreme
    T = float(rng.choice([1.0, 2.5] if moderate else [0.3, 1.0, 2.5])); weight = float(rng.choice([0.1, 1.0]))
    feat = aligned((n, d)); feat[:] = rng.normal(size=(n, d)) * float(rng.choice([0.05, 0.3] if moderate else [0.05, 0.5, 3.0]))
    if rng.random() < 0.2:
        feat[: n // 2] = feat[0]                                      # coincident features: zero distances
    lab = rng.integers(0, ncls, n).astype(np.int32)
    if tf and rng.random() < 0.5:
        lab[rng.random(n) < 0.2] = -1                                 # ignored points
    idx = rng.integers(0, n + (1 if tf else 0), (n, nsample)).astype(np.int32)   # TF: shadow entries = n
    idx[:, 0] = np.arange(n)
    per_point, mask, stats, loss = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(2, np.float32), np.zeros(1, np.float32)
    coef, own, grad = np.zeros((n, nsample), np.float32), aligned((n, d)), aligned((n, d))
    grad[:] = np.nan
    flags = (1 if tf else 0) | (4 if nce else 0) | (8 if sep else 0)
    rc = lib().cbl_contrast_pairs_forward(n, n if tf else 0x7fffffff, flags, nsample, d, P(feat), P(lab), 0, F(0.0), P(idx), None, F(T), F(weight),
                                          P(per_point), P(mask), P(stats), P(loss), P(coef), P(own), None)
    if rc != 0:
        print("CBL forward rc", it, rc, n, nsample, d); return False
    rs, rsrc = contract(idx, n)
    inv_start, inv_src = rs.astype(np.int32), np.concatenate([rsrc, np.zeros(n * nsample - len(rsrc), np.int32)]).astype(np.int32)
    one = np.ones(1, np.float32)
    rc = lib().cbl_contrast_pairs_backward(n, nsample, d, P(feat), P(coef), P(own), None, P(inv_start), P(inv_src), P(stats), P(one), F(weight), P(grad), None)
    if rc != 0:
        print("CBL backward rc", it, rc); return False
    contrast = "nce" if nce else "softnn"
    if tf:
        rl, rg, rm = C.tf_contrast(np.array(feat), lab, idx, temperature=T, weight=weight, contrast=contrast, separate=sep)
    else:
        rl, rg, rm = C.point_contrast(np.array(feat), np.eye(ncls, dtype=np.float32)[lab], idx, temperature=T, weight=weight, contrast=contrast)
    ok = np.array_equal(mask > 0, rm) and abs(float(loss[0]) - float(rl)) <= 2e-4 * max(1.0, abs(float(rl)))
    gs = max(float(np.abs(rg).max()), 1e-12)
    gerr = float(np.abs(np.array(grad) - rg).max()) / gs if rm.any() else float(np.abs(np.array(grad)).max())
    ok = ok and gerr < 5e-4
    if not ok:
        print("CBL MISMATCH", it, "tf" if tf else "pt", contrast, "S" if sep else "", n, nsample, d, ncls, T, "loss", float(loss[0]), float(rl), "gerr", gerr,
              "nan ours / oracle", int(np.isnan(np.array(grad)).sum()), int(np.isnan(rg).sum()))
        if os.environ.get("CBL_FUZZ_DUMP"):                          # the case as arrays, for a closer look
            np.savez(os.path.join(os.environ["CBL_FUZZ_DUMP"], "cbl_case_%d.npz" % it), feat=np.array(feat), lab=lab, idx=idx, T=T, weight=weight, flags=flags,
                     grad=np.array(grad), rg=rg, coef=coef, own=np.array(own), mask=mask, stats=stats, per_point=per_point)
    return ok


def _held(fn, *args):
    """a hand-written test's body on drawn parameters: True if its assertions hold"""
    try:
        fn(*args)
        return True
    except AssertionError as e:
        print("FAIL", fn.__name__, args[1:], str(e).replace("\n", " ")[:160])
        return False


def _full(names):
    L = lib()
    for name in names:
        getattr(L, name).restype = ctypes.c_size_t
    return L


def gather_case(rng, it):
    """queryandgroup / grouping / interpolation / subtraction / aggregation (tests/test_pointops_gather_host.py's checks) over point counts, neighbour counts and
    channel counts on both sides of every kernel choice: one query point, K = 1, channel counts that are not a multiple of 4, rows that start on 4-byte boundaries"""
    import tests.test_pointops_gather_host as T
    L = lib()
    which = int(rng.integers(0, 5))
    n = int(rng.choice([20, 64, 65, 300, 1000, 2100])); m = n if rng.random() < 0.5 else min(n, int(rng.choice([1, 7, 64, 150])))
    K = min(n, int(rng.choice([1, 2, 3, 8, 9, 16, 17, 32, 36])))
    c = int(rng.choice([1, 3, 4, 6, 8, 16, 32, 33, 64, 72, 128, 132]))
    if which == 0:
        return _held(T.test_queryandgroup, L, n, m, K, c, int(rng.integers(0, 2)))
    if which == 1:
        return _held(T.test_grouping, L, n, m, K, c)
    if which == 2:
        return _held(T.test_interpolation, L, max(n, 4), m, c)
    if which == 3:
        return _held(T.test_subtraction, L, n, K, c)
    c8 = int(rng.choice([8, 16, 32, 64, 128]))
    return _held(T.test_aggregation, L, n, K, c8, int(rng.choice([1, c8 // 8, c8])))


def aggregation_case(rng, it):
    """KPConv (MFMA and general kernels, scatter and gather backward), AdaptiveWeight, PosPool (tests/test_local_aggregation_host.py's checks): K = 1 .. 64 with
    shadow neighbours, widths 4 .. 288, 1 .. 16 kernel points"""
    import tests.test_local_aggregation_host as T
    L = _full(["cbl_neighbor_transpose_workspace_bytes", "cbl_adaptive_weight_backward_csr_workspace_bytes", "cbl_kpconv_backward_csr_workspace_bytes",
               "cbl_pospool_backward_csr_workspace_bytes"])
    which = int(rng.integers(0, 5))
    if which in (0, 2):
        K = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 48, 64])); C = int(rng.choice([4, 8, 12, 16, 20, 32, 64, 68, 72, 128] if which == 0 else [4, 8, 16, 32, 64, 72]))
        args = (K, C, int(rng.choice([1, 2, 7, 15, 16])), str(rng.choice(["linear", "constant"])), str(rng.choice(["sum", "closest"])))
        return _held(T.test_kpconv_forward_and_backward if which == 0 else T.test_kpconv_backward_as_a_gather, L, *args)
    if which in (1, 3):
        args = (int(rng.choice([1, 2, 9, 16, 26, 33, 41, 64])), int(rng.choice([4, 8, 16, 40, 64, 72, 144, 288])), str(rng.choice(["mean", "sum"])))
        return _held(T.test_adaptive_weight_forward_and_backward if which == 1 else T.test_adaptive_weight_backward_as_a_gather, L, *args)
    return _held(T.test_pospool_forward_and_backward, L, int(rng.choice([1, 2, 9, 16, 26, 41, 64])), int(rng.choice([6, 12, 18, 24, 36, 72])),
                 str(rng.choice(["xyz", "sin_cos"])), str(rng.choice(["mean", "sum", "max"])))


def attention_case(rng, it):
    """the attention passes and the fused layer, narrow and wide (tests/test_attention_host.py, test_pt_layer_host.py, test_pt_layer_wide_host.py): point counts that
    leave lane groups without a point, K = 1 .. 64 for the passes, every width"""
    import tests.test_attention_host as A
    import tests.test_pt_layer_host as T
    import tests.test_pt_layer_wide_host as W
    L = _full(["cbl_attn_workspace_bytes", "cbl_neighbor_transpose_workspace_bytes", "cbl_triple_linear_workspace_bytes", "cbl_pt_layer_workspace_bytes"])
    which = int(rng.integers(0, 5))
    n = int(rng.choice([9, 15, 16, 17, 63, 64, 65, 130, 400])); K = int(rng.choice([1, 2, 3, 8, 9, 16, 17, 32, 33, 64])); C = int(rng.choice([32, 64, 128, 256]))
    if which == 0:
        return _held(A.test_attention_logits_pass, L, n, K, C)
    if which == 1:
        return _held(A.test_attention_aggregation_pass, L, n, K, C, bool(rng.integers(0, 2)))
    if which == 2:
        return _held(A.test_three_projections_in_one_launch, L, int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 255, 1000])), int(rng.choice([32, 64])))
    if which == 3:
        return _held(T.test_layer_kernels_on_the_host_against_autograd, L, int(rng.choice([16, 17, 31, 33, 47, 64, 65, 100, 130])), int(rng.choice([8, 16])),
                     int(rng.choice([32, 64])), bool(rng.integers(0, 2)))
    return _held(W.test_wide_layer_on_the_host_against_autograd, L, int(rng.choice([17, 23, 40, 64, 65])), int(rng.choice([8, 16])), int(rng.choice([128, 256, 512])))


CASES = dict(knn=knn_case, radius=radius_case, grid=grid_case, subsample=subsample_case, pyramid=pyramid_case, fps=fps_case, transpose=transpose_case, cbl=cbl_case, gather=gather_case, aggregation=aggregation_case,
             attention=attention_case)


def run(which, seed, iters):
    rng = np.random.default_rng(seed)
    fails = 0
    for it in range(iters):
        try:
            ok = CASES[which](rng, it)
        except AssertionError as e:
            print("ASSERT", which, it, e); ok = False
        fails += 0 if ok else 1
    return fails


if __name__ == "__main__":
    t0 = time.time()
    which, seed, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    print(which, "seed", seed, "cases", iters, "fails", run(which, seed, iters), "%.1f s" % (time.time() - t0))
