"""CPU: the product's KPConv and AdaptiveWeight kernels (contrastboundary_amd/csrc/local_aggregation.hip: /root/reference/tensorflow/models/
local_aggregation_operators.py:620-746 and :360-484) compiled for the HOST and run with wave semantics (tests/host_emul/wave: v_mfma_f32_16x16x4_f32,
v_permlane16/32_swap, shuffles as rendezvous of a wave's fibres), through their C entry points, against the numpy restatement of the TF graph code
(oracle/local_aggregation_oracle.py) within the 1e-4 contract: the C <= 64 MFMA kernel with one and with several chunks of 16 neighbours, the general kernel,
'closest' and constant influence, shadow (padding) neighbours, the scatter-form backward passes."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import local_aggregation_oracle as LA

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "contrastboundary_amd", "csrc")
EMUL = os.path.join(HERE, "host_emul", "wave")
GEN = os.path.join(HERE, "host_emul", "host_tu.py")
TU = os.path.join(ROOT, "oracle", "_build", "local_aggregation_host.cpp")
SO = os.path.join(ROOT, "oracle", "_build", "liblocal_aggregation_host.so")


@pytest.fixture(scope="module")
def host():
    srcs = [os.path.join(CSRC, "local_aggregation.hip"), os.path.join(CSRC, "kpconv_backward.hip"), os.path.join(CSRC, "pospool.hip")]
    deps = srcs + [GEN, os.path.abspath(__file__), os.path.join(CSRC, "cbl_common.h"), os.path.join(EMUL, "amdgcn.h"), os.path.join(EMUL, "hip", "hip_runtime.h")]
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([sys.executable, GEN, TU] + srcs)
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, TU, "-o", SO])
    L = ctypes.CDLL(SO)
    L.cbl_kpconv_backward_csr_workspace_bytes.restype = ctypes.c_size_t
    L.cbl_adaptive_weight_backward_csr_workspace_bytes.restype = ctypes.c_size_t
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def aligned(a):
    """a copy of `a` whose data pointer is 16-byte aligned (feature rows are read as float4)"""
    a = np.ascontiguousarray(a)
    raw = np.zeros(a.nbytes + 16, np.uint8)
    off = (-raw.ctypes.data) % 16
    out = raw[off:off + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def make(n0, n, K, C, seed, pad_frac=0.2):
    rng = np.random.default_rng(seed)
    s = rng.uniform(0, 1, (n0, 3)).astype(np.float32)
    q = (s[rng.choice(n0, n, replace=False)] + rng.normal(0, 0.01, (n, 3))).astype(np.float32)
    d = ((s[None, :, :] - q[:, None, :]) ** 2).sum(-1)
    idx = np.argsort(d, 1)[:, :K].astype(np.int32)
    npad = rng.integers(0, int(K * pad_frac) + 1, n)
    for i in range(n):
        if npad[i]:
            idx[i, K - npad[i]:] = n0                                 # trailing shadow neighbours, as the radius search pads
    return q, s, np.ascontiguousarray(idx), aligned(rng.normal(size=(n0, C)).astype(np.float32)), rng


@pytest.mark.parametrize("K,C,KP,influence,mode", [(16, 64, 15, "linear", "sum"), (26, 64, 15, "linear", "sum"), (9, 16, 7, "linear", "closest"), (40, 32, 15, "constant", "sum"),
                                                   (26, 72, 15, "linear", "sum"), (5, 20, 16, "linear", "sum")])
def test_kpconv_forward_and_backward(host, K, C, KP, influence, mode):
    n0, n = 260, 150
    q, s, idx, f, rng = make(n0, n, K, C, seed=K + C)
    kpts = (rng.normal(size=(KP, 3)) * 0.06).astype(np.float32); kpts[0] = 0
    kw = aligned(rng.normal(size=(KP, C)).astype(np.float32))
    extent, infl, closest = 0.09, int(influence == "linear"), int(mode == "closest")
    out = aligned(np.full((n, C), np.nan, np.float32))
    assert host.cbl_kpconv_forward(n, n0, K, C, KP, P(q), P(s), P(idx), P(f), P(kpts), P(kw), ctypes.c_float(extent), infl, closest, P(out), None) == 0
    ref = LA.kpconv(q, s, idx, f, kpts, kw, extent, influence, mode)
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    go = aligned(rng.normal(size=ref.shape).astype(np.float32))
    gf, gkw = aligned(np.zeros((n0, C), np.float32)), aligned(np.zeros((KP, C), np.float32))      # the scatter form accumulates: zeroed by the caller
    assert host.cbl_kpconv_backward(n, n0, K, C, KP, P(q), P(s), P(idx), P(f), P(kpts), P(kw), ctypes.c_float(extent), infl, closest, P(go), P(gf), P(gkw), None) == 0
    rgf, rgkw = LA.kpconv_grads(q, s, idx, f, kpts, kw, extent, go, influence, mode)
    np.testing.assert_allclose(gf, rgf, rtol=1e-4, atol=1e-4 * np.abs(rgf).max())
    np.testing.assert_allclose(gkw, rgkw, rtol=1e-4, atol=1e-4 * np.abs(rgkw).max())


@pytest.mark.parametrize("K,C,reduction", [(26, 72, "mean"), (16, 64, "mean"), (21, 40, "sum"), (9, 8, "mean")])
def test_adaptive_weight_forward_and_backward(host, K, C, reduction):
    n0, n = 300, 170
    q, s, idx, f, rng = make(n0, n, K, C, seed=C)
    W = aligned((rng.normal(size=(3, C)) * 0.5).astype(np.float32)); b = aligned(rng.normal(size=(C,)).astype(np.float32))
    radius, mean = 0.1, int(reduction == "mean")
    pad = np.zeros(1, np.int32)
    assert host.cbl_index_max(ctypes.c_longlong(n * K), P(idx), P(pad), None) == 0
    assert int(pad[0]) == int(idx.max())
    out = aligned(np.full((n, C), np.nan, np.float32))
    assert host.cbl_adaptive_weight_forward(n, n0, K, C, P(q), P(s), P(idx), P(f), ctypes.c_float(radius), P(W), P(b), P(pad), mean, P(out), None) == 0
    ref = LA.adaptive_weight(q, s, idx, f, radius, W, b, reduction)
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    go = aligned(rng.normal(size=ref.shape).astype(np.float32))
    gf, gW, gb = aligned(np.zeros((n0, C), np.float32)), aligned(np.zeros((3, C), np.float32)), aligned(np.zeros(C, np.float32))
    assert host.cbl_adaptive_weight_backward(n, n0, K, C, P(q), P(s), P(idx), P(f), ctypes.c_float(radius), P(W), P(b), P(pad), mean, P(go), P(gf), P(gW), P(gb), None) == 0
    rgf, rgW, rgb = LA.adaptive_weight_grads(q, s, idx, f, radius, W, b, go, reduction)
    np.testing.assert_allclose(gf, rgf, rtol=1e-4, atol=1e-4 * np.abs(rgf).max())
    np.testing.assert_allclose(gW, rgW, rtol=1e-4, atol=1e-4 * np.abs(rgW).max())
    np.testing.assert_allclose(gb, rgb, rtol=1e-4, atol=1e-4 * np.abs(rgb).max())


def transposed_table(idx, n0):
    """cbl_neighbor_transpose's output, restated: for every target row t < n0 the ascending list of the pairs p = i * K + k whose neighbour is t (CSR: inv_start (n0 + 1),
    inv_src); shadow neighbours (== n0) are in no list"""
    flat = idx.reshape(-1)
    keep = np.nonzero(flat < n0)[0]
    order = np.argsort(flat[keep], kind="stable")
    inv_src = keep[order].astype(np.int32)
    inv_start = np.zeros(n0 + 1, np.int32)
    np.add.at(inv_start, flat[keep] + 1, 1)
    return np.cumsum(inv_start).astype(np.int32), inv_src


@pytest.mark.parametrize("K,C,KP,influence,mode", [(16, 64, 15, "linear", "sum"), (26, 72, 15, "linear", "sum"), (9, 16, 7, "linear", "closest"), (40, 32, 15, "constant", "sum")])
def test_kpconv_backward_as_a_gather(host, K, C, KP, influence, mode):
    """cbl_kpconv_backward_csr (csrc/kpconv_backward.hip: the scatter-add of the feature gradient as a gather over the transposed neighbour table, on MFMA) — the
    backward pass the headline step runs — against the same analytic gradients; outputs are WRITTEN (pre-filled with NaN here), no atomics"""
    n0, n = 260, 150
    q, s, idx, f, rng = make(n0, n, K, C, seed=3 * K + C)
    kpts = (rng.normal(size=(KP, 3)) * 0.06).astype(np.float32); kpts[0] = 0
    kw = aligned(rng.normal(size=(KP, C)).astype(np.float32))
    extent, infl, closest = 0.09, int(influence == "linear"), int(mode == "closest")
    go = aligned(rng.normal(size=(n, C)).astype(np.float32))
    inv_start, inv_src = transposed_table(idx, n0)
    gf, gkw = aligned(np.full((n0, C), np.nan, np.float32)), aligned(np.full((KP, C), np.nan, np.float32))
    nbytes = host.cbl_kpconv_backward_csr_workspace_bytes(n0, C, KP)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = host.cbl_kpconv_backward_csr(n, n0, K, C, KP, P(q), P(s), P(f), P(kpts), P(kw), ctypes.c_float(extent), infl, closest, P(go), None, P(inv_start), P(inv_src),
                                      P(gf), P(gkw), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    rgf, rgkw = LA.kpconv_grads(q, s, idx, f, kpts, kw, extent, go, influence, mode)
    np.testing.assert_allclose(gf, rgf, rtol=1e-4, atol=1e-4 * np.abs(rgf).max())
    np.testing.assert_allclose(gkw, rgkw, rtol=1e-4, atol=1e-4 * np.abs(rgkw).max())


@pytest.mark.parametrize("K,C,reduction", [(26, 72, "mean"), (16, 64, "mean"), (21, 40, "sum"), (9, 8, "mean"), (12, 288, "mean")])
def test_adaptive_weight_backward_as_a_gather(host, K, C, reduction):
    """cbl_adaptive_weight_backward_csr — the backward pass of the ConvNet step (cbl_convnet_step) — over the same numpy-built table: every lane width
    (lanes own one, two or three float4 columns), outputs written over NaN, no atomics"""
    n0, n = 300, 170
    q, s, idx, f, rng = make(n0, n, K, C, seed=7 * C + K)
    W = aligned((rng.normal(size=(3, C)) * 0.5).astype(np.float32)); b = aligned(rng.normal(size=(C,)).astype(np.float32))
    radius, mean = 0.1, int(reduction == "mean")
    pad = np.array([int(idx.max())], np.int32)
    go = aligned(rng.normal(size=(n, C)).astype(np.float32))
    inv_start, inv_src = transposed_table(idx, n0)
    gf, gW, gb = aligned(np.full((n0, C), np.nan, np.float32)), aligned(np.full((3, C), np.nan, np.float32)), aligned(np.full(C, np.nan, np.float32))
    nbytes = host.cbl_adaptive_weight_backward_csr_workspace_bytes(n, n0, C)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = host.cbl_adaptive_weight_backward_csr(n, n0, K, C, P(q), P(s), P(idx), P(f), ctypes.c_float(radius), P(W), P(b), P(pad), mean, P(go), None, P(inv_start), P(inv_src),
                                               P(gf), P(gW), P(gb), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    rgf, rgW, rgb = LA.adaptive_weight_grads(q, s, idx, f, radius, W, b, go, reduction)
    np.testing.assert_allclose(gf, rgf, rtol=1e-4, atol=1e-4 * np.abs(rgf).max())
    np.testing.assert_allclose(gW, rgW, rtol=1e-4, atol=1e-4 * np.abs(rgW).max())
    np.testing.assert_allclose(gb, rgb, rtol=1e-4, atol=1e-4 * np.abs(rgb).max())


POSPOOL_EMBEDDINGS = {"one": 0, "xyz": 1, "distance": 2, "exp_-d": 3, "direction_exp_-d": 4, "direction_d": 5, "sin_cos": 6, "two_order": 7, "three_order": 8}
POSPOOL_REDUCTIONS = {"sum": 0, "mean": 1, "max": 2}


@pytest.mark.parametrize("K,C,embedding,reduction", [(16, 72, "sin_cos", "mean"), (20, 36, "xyz", "sum"), (9, 9, "sin_cos", "mean"), (26, 18, "two_order", "mean"),
                                                     (12, 36, "direction_exp_-d", "mean"), (16, 24, "distance", "max")])
def test_pospool_forward_and_backward(host, K, C, embedding, reduction):
    """PosPool (csrc/pospool.hip; local_aggregation_operators.py:15-250): parameter-free position embeddings (v_sin_f32 works in revolutions) times the gathered
    features, reduced over the neighbours; the gradient of the features through the scatter entry"""
    n0, n = 280, 160
    q, s, idx, f, rng = make(n0, n, K, C, seed=C + 13 * K)
    radius = 0.1
    pad = np.array([int(idx.max())], np.int32)
    out = aligned(np.full((n, C), np.nan, np.float32))
    pe, red = POSPOOL_EMBEDDINGS[embedding], POSPOOL_REDUCTIONS[reduction]
    assert host.cbl_pospool_forward(n, n0, K, C, P(q), P(s), P(idx), P(f), ctypes.c_float(radius), pe, red, P(pad), P(out), None) == 0
    ref, _, _ = LA.pospool(q, s, idx, f, radius, embedding, reduction)
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    if reduction == "max":
        return
    go = aligned(rng.normal(size=ref.shape).astype(np.float32))
    gf = aligned(np.zeros((n0, C), np.float32))
    assert host.cbl_pospool_backward(n, n0, K, C, P(q), P(s), P(idx), P(f), ctypes.c_float(radius), pe, red, P(pad), P(go), P(gf), None) == 0
    rgf = LA.pospool_grad_features(q, s, idx, f, radius, go, embedding, reduction)
    np.testing.assert_allclose(gf, rgf, rtol=1e-4, atol=1e-4 * np.abs(rgf).max())
