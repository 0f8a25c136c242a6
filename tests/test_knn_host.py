"""CPU: the product's exact K-nearest-neighbour search (contrastboundary_amd/csrc/knn_grid.hip: cell grid + group / wave kernels with certification,
knn_select.hip: block select, knn_exact.hip: brute force and the reference-order replay of tied rows, knn_dispatch.hip: the entry points) compiled for the HOST and
run with wave semantics (tests/host_emul/wave), through `cbl_knnquery` and friends, against the oracle's restatement of knnquery_cuda_kernel
(/root/reference/pytorch/lib/pointops/src/knnquery/knnquery_cuda_kernel.cu:65-111): indices AND squared distances bit for bit — on uniform clouds, surface-like
clouds, lattices (every row tied: the replay), ragged batches, queries that are not the supports, every kernel family (K <= 16: groups of lanes; K = 36: one wave per
query, select-then-sort; the nested K = 36 / 16 search the bench step runs; brute force)."""
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
BUILD = os.path.join(ROOT, "oracle", "_build")
SO = os.path.join(BUILD, "libknn_host.so")
FILES = ["knn_exact", "knn_select", "knn_grid", "knn_dispatch"]


@pytest.fixture(scope="module")
def host():
    srcs = [os.path.join(CSRC, f + ".hip") for f in FILES]
    deps = srcs + [GEN, os.path.abspath(__file__), os.path.join(CSRC, "cbl_common.h"), os.path.join(CSRC, "grid_core.h"), os.path.join(EMUL, "amdgcn.h"),
                   os.path.join(EMUL, "hip", "hip_runtime.h")]
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        objs = []
        for f, src in zip(FILES, srcs):                               # one translation unit per file, as in the product build (their anonymous namespaces overlap)
            tu, obj = os.path.join(BUILD, f + "_host.cpp"), os.path.join(BUILD, f + "_host.o")
            subprocess.check_call([sys.executable, GEN, tu, src])
            subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-c", "-ffp-contract=off", "-Wno-unknown-pragmas",
                                   "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, tu, "-o", obj])
            objs.append(obj)
        subprocess.check_call(["g++", "-shared"] + objs + ["-o", SO])
    L = ctypes.CDLL(SO)
    L.cbl_knnquery_workspace_bytes.restype = ctypes.c_size_t
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def cloud(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(0, 1, (n, 3)).astype(np.float32)
    if kind == "surface":
        u = rng.uniform(0, 1, (n, 2)).astype(np.float32)
        z = np.where(np.arange(n) % 2 == 0, 0.0, u[:, 0] * 0.3).astype(np.float32)
        return np.concatenate([u, z[:, None]], 1)
    if kind == "grid":                                                # a whole lattice: every row is tied at its K-th distance and takes the replay
        s = int(round(n ** (1 / 3.0))) + 1
        g = np.stack(np.meshgrid(np.arange(s), np.arange(s), np.arange(s), indexing="ij"), -1).reshape(-1, 3)[:n].astype(np.float32) * 0.0625
        return g[rng.permutation(n)]
    if kind == "lattice":                                             # a uniform cloud with a lattice patch inside: a few dozen rows whose distances are exactly tied take
        s = 3                                                         # the tie logic and the replay
        g = np.stack(np.meshgrid(np.arange(s), np.arange(s), np.arange(s), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * 0.0625 + 0.3
        pts = np.concatenate([rng.uniform(0, 1, (n - len(g), 3)).astype(np.float32), g])
        return pts[rng.permutation(n)]
    raise ValueError(kind)


def run(L, entry, K, xyz, q, off, qoff):
    xyz, q, off, qoff = O.f32(xyz), O.f32(q), O.i32(off), O.i32(qoff)
    b, n, m = len(off), xyz.shape[0], q.shape[0]
    idx, d2 = np.full((m, K), -7, np.int32), np.full((m, K), np.nan, np.float32)
    nbytes = L.cbl_knnquery_workspace_bytes(b, n, m, K)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = getattr(L, entry)(b, n, m, K, P(xyz), P(q), P(off), P(qoff), P(idx), P(d2), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0, rc
    return idx, d2


@pytest.mark.parametrize("kind,sizes,K", [("uniform", [2300], 16), ("uniform", [2300], 36), ("surface", [2500], 8), ("lattice", [2200], 16),
                                          ("lattice", [2100], 36), ("uniform", [2100, 300], 16),
                                          pytest.param("grid", [2197], 16, marks=pytest.mark.skipif(not os.environ.get("CBL_HOST_EMUL_FULL"), reason="every row replayed: 16 s; set CBL_HOST_EMUL_FULL=1"))])
def test_self_queries_equal_the_oracle_bit_for_bit(host, kind, sizes, K):
    xyz = np.concatenate([cloud(kind, n, 20 + i) + 2.5 * i for i, n in enumerate(sizes)])
    off = np.cumsum(sizes)
    idx, d2 = run(host, "cbl_knnquery", K, xyz, xyz, off, off)
    ridx, rd2 = O.knnquery(K, xyz, xyz, off, off)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(d2.view(np.uint32), rd2.view(np.uint32))


def test_foreign_queries_and_the_brute_force_entry(host):
    xyz = cloud("uniform", 2400, 3)
    q = np.concatenate([cloud("uniform", 200, 4), xyz[:50]])          # some queries coincide with supports
    off, qoff = np.int32([1100, 2400]), np.int32([120, 250])
    ridx, rd2 = O.knnquery(16, xyz, q, off, qoff)
    idx, d2 = run(host, "cbl_knnquery", 16, xyz, q, off, qoff)
    np.testing.assert_array_equal(idx, ridx); np.testing.assert_array_equal(d2.view(np.uint32), rd2.view(np.uint32))
    idx2, d22 = np.zeros_like(ridx), np.zeros_like(rd2)
    assert host.cbl_knnquery_exact(2, 2400, 250, 16, P(xyz), P(q), P(off), P(qoff), P(idx2), P(d22), None) == 0
    np.testing.assert_array_equal(idx2, ridx); np.testing.assert_array_equal(d22.view(np.uint32), rd2.view(np.uint32))


@pytest.mark.parametrize("kind", ["uniform", "lattice"])
def test_nested_search_equals_two_searches(host, kind):
    """cbl_knnquery_nested: the K = 36 search of the CBL head with the K = 16 table of the block derived from it (what the bench step runs)"""
    n = 2300
    xyz = cloud(kind, n, 9)
    off = np.int32([n])
    iw, dw = np.zeros((n, 36), np.int32), np.zeros((n, 36), np.float32)
    i16, d16 = np.zeros((n, 16), np.int32), np.zeros((n, 16), np.float32)
    nbytes = host.cbl_knnquery_workspace_bytes(1, n, n, 36)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = host.cbl_knnquery_nested(1, n, n, 36, 0, 16, 0, P(xyz), P(xyz), P(off), P(off), P(iw), P(dw), P(i16), P(d16), None, None, P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    r36, rd36 = O.knnquery(36, xyz, xyz, off, off)
    r16, rd16 = O.knnquery(16, xyz, xyz, off, off)
    np.testing.assert_array_equal(iw, r36); np.testing.assert_array_equal(dw.view(np.uint32), rd36.view(np.uint32))
    np.testing.assert_array_equal(i16, r16); np.testing.assert_array_equal(d16.view(np.uint32), rd16.view(np.uint32))
