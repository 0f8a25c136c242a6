"""CPU: the product's neighbour-table transposition (contrastboundary_amd/csrc/neighbor_transpose.hip — the table behind every gather-form backward pass of the
path: K4 grouping_cuda_kernel.cu:16-25, the CBL gradient, KPConv / AdaptiveWeight / attention backward) compiled for the HOST and run with wave semantics
(tests/host_emul/wave), through `cbl_neighbor_transpose` and the consumers defined beside it, against its contract written in numpy: segment r of inv_src lists,
ASCENDING, the flat pair indices p = source * nsample + column with idx[p] == order_dst[r]; shadow entries are in no list.  The table bit for bit — hub targets
(lists longer than a wave), targets nobody lists, a processing order, K on both sides of the per-thread batch sizes — and K4 over it against the oracle."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "contrastboundary_amd", "csrc")
EMUL = os.path.join(HERE, "host_emul", "wave")
GEN = os.path.join(HERE, "host_emul", "host_tu.py")
TU = os.path.join(ROOT, "oracle", "_build", "neighbor_transpose_host.cpp")
SO = os.path.join(ROOT, "oracle", "_build", "libneighbor_transpose_host.so")


@pytest.fixture(scope="module")
def host():
    srcs = [os.path.join(CSRC, "neighbor_transpose.hip")]
    deps = srcs + [GEN, os.path.abspath(__file__), os.path.join(CSRC, "cbl_common.h"), os.path.join(CSRC, "k4_rows_pipe.h"), os.path.join(EMUL, "gather_wave.h"),
                   os.path.join(EMUL, "amdgcn.h"), os.path.join(EMUL, "hip", "hip_runtime.h")]
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([sys.executable, GEN, TU] + srcs)
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, TU, "-o", SO])
    L = ctypes.CDLL(SO)
    L.cbl_neighbor_transpose_workspace_bytes.restype = ctypes.c_size_t
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def contract(idx, n, order_dst=None):
    flat = idx.reshape(-1)
    keep = np.nonzero((flat >= 0) & (flat < n))[0]
    tgt = flat[keep]
    if order_dst is not None:                                        # segment r belongs to target order_dst[r]
        pos = np.empty(n, np.int64); pos[order_dst] = np.arange(n)
        tgt = pos[tgt]
    o = np.argsort(tgt, kind="stable")
    inv_start = np.zeros(n + 1, np.int64)
    np.add.at(inv_start, tgt + 1, 1)
    return np.cumsum(inv_start).astype(np.int32), keep[o].astype(np.int32)


def build(L, idx, n, order=None):
    m, K = idx.shape
    inv_start, inv_src = np.full(n + 1, -1, np.int32), np.full(m * K, -1, np.int32)
    nbytes = L.cbl_neighbor_transpose_workspace_bytes(m, n, K)
    ws = np.zeros(nbytes + 64, np.uint8)
    assert L.cbl_neighbor_transpose(m, n, K, P(idx), P(order), P(order), P(inv_start), P(inv_src), P(ws), ctypes.c_size_t(nbytes), None) == 0
    return inv_start, inv_src


def knn_table(n, m, K, seed):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    q = xyz if m == n else xyz[rng.choice(n, m, replace=False)]
    idx, _ = O.knnquery(K, xyz, np.ascontiguousarray(q), np.int32([n]), np.int32([m]))
    return np.ascontiguousarray(idx, np.int32), rng


@pytest.mark.parametrize("n,m,K,kind", [(900, 900, 16, "knn"), (700, 700, 36, "knn"), (800, 300, 8, "knn"), (600, 600, 16, "hub"), (500, 500, 9, "shadow"), (1500, 1500, 4, "knn")])
def test_table_equals_its_contract(host, n, m, K, kind):
    idx, rng = knn_table(n, m, K, seed=n + K)
    if kind == "hub":
        idx[:, 3] = 7                                                # every source lists target 7: one list of m entries, far longer than a wave
        idx[::2, 5] = 11
    if kind == "shadow":
        idx[rng.uniform(size=idx.shape) < 0.2] = n                   # the radius search's padding: in no list
    inv_start, inv_src = build(host, idx, n)
    rs, rsrc = contract(idx, n)
    np.testing.assert_array_equal(inv_start, rs)
    np.testing.assert_array_equal(inv_src[:rs[-1]], rsrc)


def test_table_under_a_processing_order_and_k4_over_it(host):
    """order_dst permutes the SEGMENTS (segment r = target order[r]); K4 as a gather over the table (cbl_grouping_backward_csr) equals the oracle's scatter-add"""
    n, K, c = 800, 16, 32
    idx, rng = knn_table(n, n, K, seed=5)
    order = rng.permutation(n).astype(np.int32)
    inv_start, inv_src = build(host, idx, n, order)
    rs, rsrc = contract(idx, n, order)
    np.testing.assert_array_equal(inv_start, rs)
    np.testing.assert_array_equal(inv_src[:rs[-1]], rsrc)
    go = rng.normal(size=(n, K, c)).astype(np.float32)
    raw = np.zeros(n * c * 4 + 16, np.uint8); off = (-raw.ctypes.data) % 16
    gi = raw[off:off + n * c * 4].view(np.float32).reshape(n, c); gi[:] = np.nan
    raw2 = np.zeros(go.nbytes + 16, np.uint8); off2 = (-raw2.ctypes.data) % 16
    goa = raw2[off2:off2 + go.nbytes].view(np.float32).reshape(go.shape); goa[:] = go
    assert host.cbl_grouping_backward_csr(n, c, P(goa), P(order), P(inv_start), P(inv_src), P(gi), None) == 0
    ref = O.grouping_backward(go, idx, n)
    np.testing.assert_allclose(gi, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
