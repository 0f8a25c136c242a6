"""CPU: the product's neighbour gather / scatter operators (contrastboundary_amd/csrc/pointops_gather.hip: queryandgroup — the north-star kernel —, grouping,
interpolation, subtraction, aggregation; /root/reference/pytorch/lib/pointops/src/{grouping,interpolation,subtraction,aggregation}/*_cuda_kernel.cu and
functions/pointops.py:95-127) compiled for the HOST and run with wave semantics (tests/host_emul/wave), through their C entry points, against the oracle
(oracle/pointops_oracle.c, pinned by the reference kernels' own outputs: tests/golden/pointops_*.npz): forward passes bit for bit (they copy, subtract or accumulate
in the reference's order), scatter-form backward passes within the order of float atomics."""
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
TU = os.path.join(ROOT, "oracle", "_build", "pointops_gather_host.cpp")
SO = os.path.join(ROOT, "oracle", "_build", "libpointops_gather_host.so")


@pytest.fixture(scope="module")
def host():
    srcs = [os.path.join(CSRC, "pointops_gather.hip")]
    deps = srcs + [GEN, os.path.abspath(__file__), os.path.join(CSRC, "cbl_common.h"), os.path.join(CSRC, "query_group_pipe.h"), os.path.join(EMUL, "gather_wave.h"),
                   os.path.join(EMUL, "amdgcn.h"), os.path.join(EMUL, "hip", "hip_runtime.h")]
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([sys.executable, GEN, TU] + srcs)
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, TU, "-o", SO])
    return ctypes.CDLL(SO)


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def aligned(a):
    a = np.ascontiguousarray(a)
    raw = np.zeros(a.nbytes + 16, np.uint8)
    off = (-raw.ctypes.data) % 16
    out = raw[off:off + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def scene(n, m, K, c, seed):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    q = xyz[rng.choice(n, m, replace=False)] if m < n else xyz
    off, qoff = np.int32([n]), np.int32([m])
    idx, _ = O.knnquery(K, xyz, q, off, qoff)
    return xyz, np.ascontiguousarray(q), aligned(rng.normal(size=(n, c)).astype(np.float32)), np.ascontiguousarray(idx, np.int32), rng


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("n,m,K,c,use_xyz", [(700, 700, 16, 64, 1), (600, 150, 8, 32, 1), (500, 500, 16, 13, 1), (400, 400, 9, 64, 0), (300, 300, 16, 3, 1), (513, 513, 36, 32, 1)])
def test_queryandgroup(host, n, m, K, c, use_xyz):
    """out[i, k] = [xyz[idx[i, k]] - new_xyz[i] | feat[idx[i, k]]]  (pointops.py:95-127): bits"""
    xyz, q, feat, idx, _ = scene(n, m, K, c, seed=n + c)
    w = c + (3 if use_xyz else 0)
    out = aligned(np.full((m, K, w), np.nan, np.float32))
    assert host.cbl_queryandgroup(m, K, c, use_xyz, P(xyz), P(q), P(feat), P(idx), P(out), None) == 0
    ref = feat[idx]
    if use_xyz:
        ref = np.concatenate([xyz[idx] - q[:, None, :], ref], -1)
    np.testing.assert_array_equal(bits(out), bits(ref.astype(np.float32)))


@pytest.mark.parametrize("n,m,K,c", [(600, 200, 16, 64), (500, 500, 8, 20), (300, 300, 5, 7)])
def test_grouping(host, n, m, K, c):
    xyz, q, feat, idx, rng = scene(n, m, K, c, seed=3 * n + c)
    out = aligned(np.full((m, K, c), np.nan, np.float32))
    assert host.cbl_grouping_forward(m, K, c, P(feat), P(idx), P(out), None) == 0
    np.testing.assert_array_equal(bits(out), bits(O.grouping_forward(feat, idx)))
    go = aligned(rng.normal(size=(m, K, c)).astype(np.float32))
    gi = aligned(np.zeros((n, c), np.float32))
    assert host.cbl_grouping_backward(m, K, c, P(go), P(idx), P(gi), None) == 0
    ref = O.grouping_backward(go, idx, n)
    np.testing.assert_allclose(gi, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())


@pytest.mark.parametrize("n,m,c", [(500, 150, 64), (400, 100, 20)])
def test_interpolation(host, n, m, c):
    """n fine points take the inverse-distance mean of their 3 nearest coarse points (pointops.py:167-210)"""
    rng = np.random.default_rng(n + c)
    fine, coarse = rng.uniform(0, 1, (n, 3)).astype(np.float32), rng.uniform(0, 1, (m, 3)).astype(np.float32)
    idx, d2 = O.knnquery(3, coarse, fine, np.int32([m]), np.int32([n]))
    w = 1.0 / (np.sqrt(d2) + 1e-8); w = (w / w.sum(1, keepdims=True)).astype(np.float32)
    inp = aligned(rng.normal(size=(m, c)).astype(np.float32))
    out = aligned(np.zeros((n, c), np.float32))                        # accumulated into (interpolation_cuda_kernel.cu: atomicAdd into a zeroed output)
    assert host.cbl_interpolation_forward(n, c, 3, P(inp), P(idx), P(w), P(out), None) == 0
    ref = O.interpolation_forward(inp, idx, w)
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-6 * np.abs(ref).max())
    go = aligned(rng.normal(size=(n, c)).astype(np.float32))
    gi = aligned(np.zeros((m, c), np.float32))
    assert host.cbl_interpolation_backward(n, c, 3, P(go), P(idx), P(w), P(gi), None) == 0
    rgi = O.interpolation_backward(go, idx, w, m)
    np.testing.assert_allclose(gi, rgi, rtol=1e-5, atol=1e-5 * np.abs(rgi).max())


@pytest.mark.parametrize("n,K,c", [(500, 16, 64), (300, 8, 20)])
def test_subtraction(host, n, K, c):
    xyz, q, a, idx, rng = scene(n, n, K, c, seed=5 * n + c)
    b = aligned(rng.normal(size=(n, c)).astype(np.float32))
    out = aligned(np.full((n, K, c), np.nan, np.float32))
    assert host.cbl_subtraction_forward(n, K, c, P(a), P(b), P(idx), P(out), None) == 0
    np.testing.assert_array_equal(bits(out), bits(O.subtraction_forward(a, b, idx)))
    go = aligned(rng.normal(size=(n, K, c)).astype(np.float32))
    g1, g2 = aligned(np.zeros((n, c), np.float32)), aligned(np.zeros((n, c), np.float32))
    assert host.cbl_subtraction_backward(n, K, c, P(idx), P(go), P(g1), P(g2), None) == 0
    r1, r2 = O.subtraction_backward(idx, go)
    np.testing.assert_allclose(g1, r1, rtol=1e-5, atol=1e-5 * np.abs(r1).max())
    np.testing.assert_allclose(g2, r2, rtol=1e-5, atol=1e-5 * np.abs(r2).max())


@pytest.mark.parametrize("n,K,c,wc", [(400, 16, 64, 8), (300, 8, 32, 4), (200, 16, 24, 3)])
def test_aggregation(host, n, K, c, wc):
    """out[i, c] = sum_k (input[idx[i, k], c] + position[i, k, c]) * weight[i, k, c % w_c]  (aggregation_cuda_kernel.cu)"""
    xyz, q, inp, idx, rng = scene(n, n, K, c, seed=7 * n + c)
    pos = aligned(rng.normal(size=(n, K, c)).astype(np.float32)); w = aligned(rng.normal(size=(n, K, wc)).astype(np.float32))
    out = aligned(np.zeros((n, c), np.float32))                        # accumulated into, like the reference's zeroed output
    assert host.cbl_aggregation_forward(n, K, c, wc, P(inp), P(pos), P(w), P(idx), P(out), None) == 0
    ref = O.aggregation_forward(inp, pos, w, idx)
    np.testing.assert_allclose(out, ref, rtol=1e-6, atol=1e-6 * np.abs(ref).max())
    go = aligned(rng.normal(size=(n, c)).astype(np.float32))
    gi, gp, gw = aligned(np.zeros((n, c), np.float32)), aligned(np.zeros((n, K, c), np.float32)), aligned(np.zeros((n, K, wc), np.float32))
    assert host.cbl_aggregation_backward(n, K, c, wc, P(inp), P(pos), P(w), P(idx), P(go), P(gi), P(gp), P(gw), None) == 0
    rgi, rgp, rgw = O.aggregation_backward(inp, pos, w, idx, go)
    np.testing.assert_allclose(gi, rgi, rtol=1e-5, atol=1e-5 * np.abs(rgi).max())
    np.testing.assert_allclose(gp, rgp, rtol=1e-6, atol=1e-6 * np.abs(rgp).max())
    np.testing.assert_allclose(gw, rgw, rtol=1e-5, atol=1e-5 * np.abs(rgw).max())
