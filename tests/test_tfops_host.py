"""CPU: the product's TF-side operators (contrastboundary_amd/csrc/tfops.hip over the grid of knn_grid.hip: grid subsampling —
/root/reference/tensorflow/ops/cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp — and the sorted radius search cropped to a limit —
tf_custom_ops/tf_neighbors/neighbors/neighbors.cpp:213-336, datasets/base.py:756-765) compiled for the HOST and run with wave semantics (tests/host_emul/wave),
through `cbl_grid_subsampling` / `cbl_radius_neighbors`, against the oracle (oracle/tfops_oracle.c, itself pinned by the reference's own C++ built as oracle/_ref):
barycentres, voxel counts, features and majority labels bit for bit in the canonical order; neighbour tables, counts and the largest neighbourhood bit for bit.
The whole input pyramid as one native call (pyramid.hip: `cbl_pyramid`, datasets/base.py:767-842) against the oracle's operators applied layer by layer."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from contrastboundary_amd import synthetic as S
from tests import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "contrastboundary_amd", "csrc")
EMUL = os.path.join(HERE, "host_emul", "wave")
GEN = os.path.join(HERE, "host_emul", "host_tu.py")
BUILD = os.path.join(ROOT, "oracle", "_build")
SO = os.path.join(BUILD, "libtfops_host.so")
FILES = ["tfops", "pyramid", "knn_grid", "knn_exact", "knn_select", "knn_dispatch"]


@pytest.fixture(scope="module")
def host():
    srcs = [os.path.join(CSRC, f + ".hip") for f in FILES]
    deps = srcs + [GEN, os.path.abspath(__file__), os.path.join(CSRC, "cbl_common.h"), os.path.join(CSRC, "grid_core.h"), os.path.join(EMUL, "amdgcn.h"),
                   os.path.join(EMUL, "hip", "hip_runtime.h"), os.path.join(EMUL, "rocprim", "device", "device_radix_sort.hpp")]
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        objs = []
        for f, src in zip(FILES, srcs):
            tu, obj = os.path.join(BUILD, f + "_host.cpp"), os.path.join(BUILD, f + "_tfhost.o")
            subprocess.check_call([sys.executable, GEN, tu, src])
            subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-c", "-ffp-contract=off", "-Wno-unknown-pragmas",
                                   "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, tu, "-o", obj])
            objs.append(obj)
        subprocess.check_call(["g++", "-shared"] + objs + ["-o", SO])
    L = ctypes.CDLL(SO)
    L.cbl_grid_subsampling_workspace_bytes.restype = ctypes.c_size_t
    L.cbl_radius_neighbors_workspace_bytes.restype = ctypes.c_size_t
    L.cbl_pyramid_layer_workspace_bytes.restype = ctypes.c_size_t
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("dl", [0.08, 0.3])
def test_grid_subsampling(host, dl):
    xyz, lab = S.s_room(4000, seed=3)
    lens = np.int32([1400, 1, 2599]); off = np.cumsum(lens).astype(np.int32)
    n, b = xyz.shape[0], len(lens)
    rng = np.random.default_rng(1)
    feat = rng.uniform(size=(n, 3)).astype(np.float32)
    labels = np.stack([lab, rng.integers(0, 3, n)], 1).astype(np.int32)
    op, of, ol = np.full((n, 3), np.nan, np.float32), np.full((n, 3), np.nan, np.float32), np.full((n, 2), -1, np.int32)
    out_len, total = np.full(b, -1, np.int32), np.full(1, -1, np.int32)
    nbytes = host.cbl_grid_subsampling_workspace_bytes(b, n)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = host.cbl_grid_subsampling(b, n, P(xyz), P(off), ctypes.c_float(dl), 3, P(feat), 2, P(labels), P(op), P(of), P(ol), P(out_len), P(total), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    rp, rl = O.grid_subsampling(xyz, lens, dl)                      # the batched flavour: points and per-cloud voxel counts
    m = int(total[0])
    np.testing.assert_array_equal(out_len, rl)
    assert m == rp.shape[0]
    np.testing.assert_array_equal(op[:m].view(np.uint32), rp.view(np.uint32))
    # features / majority labels: the wrapper flavour works on one cloud; check cloud by cloud
    s = 0; t = 0
    for c in range(b):
        e = s + int(lens[c])
        fp, ff, fl, _ = O.grid_subsampling_full(xyz[s:e], feat[s:e], labels[s:e], dl)
        k = fp.shape[0]
        np.testing.assert_array_equal(of[t:t + k].view(np.uint32), ff.view(np.uint32))
        np.testing.assert_array_equal(ol[t:t + k], fl)
        s, t = e, t + k


@pytest.mark.parametrize("r,limit", [(0.2, 31), (0.1, 8), (0.3, 64)])
def test_radius_neighbors(host, r, limit):
    xyz, _ = S.s_room(3000, seed=5)
    lens = np.int32([1200, 1800])
    sub = np.concatenate([xyz[:1200:3], xyz[1200::3]]); sl = np.int32([len(xyz[:1200:3]), len(xyz[1200::3])])
    rng = np.random.default_rng(0)
    far = np.concatenate([xyz[:40] + 0.03, rng.uniform(20, 21, (5, 3)).astype(np.float32), xyz[1200:1240] - 0.02]).astype(np.float32)
    for (q, ql, s, slen) in [(xyz, lens, xyz, lens), (sub, sl, xyz, lens), (xyz, lens, sub, sl), (far, np.int32([45, 40]), xyz, lens)]:
        q, s = np.ascontiguousarray(q, np.float32), np.ascontiguousarray(s, np.float32)
        b, nq, ns = len(ql), q.shape[0], s.shape[0]
        qo, so = np.cumsum(ql).astype(np.int32), np.cumsum(slen).astype(np.int32)
        out, counts, mx = np.full((nq, limit), -1, np.int32), np.full(nq, -1, np.int32), np.zeros(1, np.int32)
        nbytes = host.cbl_radius_neighbors_workspace_bytes(b, ns)
        ws = np.zeros(nbytes + 64, np.uint8)
        rc = host.cbl_radius_neighbors(b, nq, ns, P(q), P(s), P(qo), P(so), ctypes.c_float(r), limit, P(out), P(counts), P(mx), P(ws), ctypes.c_size_t(nbytes), None)
        assert rc == 0
        ref, rcounts, mc = O.radius_neighbors(q, s, ql, slen, r, limit)
        np.testing.assert_array_equal(out, ref)
        np.testing.assert_array_equal(counts, rcounts)
        assert int(mx[0]) == mc


def test_pyramid_in_one_call_equals_the_operators_layer_by_layer(host):
    """tf_segmentation_inputs_radius (datasets/base.py:795-820): per layer conv_i = neighbors(points, points, r) cropped to the layer's limit, pool points =
    subsampling at 2 dl, pool_i = neighbors(pool, points, r), up_i = neighbors(points, pool, 2 r); r and dl double from layer to layer; the last layer only
    searches.  Layer l > 0 reuses the search grid its predecessor's upsampling built (grid_is_built), which this case exercises twice."""
    xyz, _ = S.s_room(3600, seed=11)
    lens = np.int32([1500, 2100])
    b, n = len(lens), xyz.shape[0]
    r0, dl0, layers = 0.12, 0.05, 3
    limits = np.int32([24, 20, 33])
    grid_bytes = host.cbl_radius_neighbors_workspace_bytes(b, n)
    grids = [np.zeros(grid_bytes + 64, np.uint8) for _ in range(layers)]
    nb = [np.full((n, int(limits[l])), -5, np.int32) for l in range(layers)]
    pp = [np.full((n, 3), np.nan, np.float32) for _ in range(layers - 1)]
    pl = [np.full(b, -1, np.int32) for _ in range(layers - 1)]
    po = [np.full((n, int(limits[l])), -5, np.int32) for l in range(layers - 1)]
    up = [np.full((n, int(limits[l])), -5, np.int32) for l in range(layers - 1)]
    mx, sizes = np.full(3 * layers, -1, np.int32), np.full(layers, -1, np.int32)
    nbytes = host.cbl_pyramid_layer_workspace_bytes(b, n)
    ws = np.zeros(nbytes + 64, np.uint8)

    def ptrs(arrs, count):
        a = (ctypes.c_void_p * count)()
        for i, x in enumerate(arrs):
            a[i] = x.ctypes.data
        return a
    rc = host.cbl_pyramid(b, n, P(xyz), P(lens), ctypes.c_float(r0), ctypes.c_float(dl0), layers, P(limits), ptrs(grids, layers), ctypes.c_size_t(grid_bytes),
                          ptrs(nb, layers), ptrs(pp, layers), ptrs(pl, layers), ptrs(po, layers), ptrs(up, layers), P(mx), P(sizes), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    pts, ln, r, dl = xyz, lens, r0, dl0
    for l in range(layers):
        m = pts.shape[0]
        assert int(sizes[l]) == m
        lim = int(limits[l])
        ref, _, mc = O.radius_neighbors(pts, pts, ln, ln, r, lim)
        np.testing.assert_array_equal(nb[l][:m], ref)
        assert int(mx[3 * l]) == mc
        if l == layers - 1:
            break
        sub, sl = O.grid_subsampling(pts, ln, 2 * dl)
        k = sub.shape[0]
        assert 0 < k < m
        np.testing.assert_array_equal(pl[l], sl)
        np.testing.assert_array_equal(pp[l][:k].view(np.uint32), sub.view(np.uint32))
        ref, _, mc = O.radius_neighbors(sub, pts, sl, ln, r, lim)
        np.testing.assert_array_equal(po[l][:k], ref); assert int(mx[3 * l + 1]) == mc
        ref, _, mc = O.radius_neighbors(pts, sub, ln, sl, 2 * r, lim)
        np.testing.assert_array_equal(up[l][:m], ref); assert int(mx[3 * l + 2]) == mc
        pts, ln, r, dl = np.ascontiguousarray(sub), sl.astype(np.int32), 2 * r, 2 * dl
