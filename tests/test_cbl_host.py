"""CPU: the product's CBL pair-mining kernels (contrastboundary_amd/csrc/cbl_pairs.hip + the finalize pass of cbl.hip: positives / negatives among a point's
neighbours, soft-nearest-neighbour or NCE loss, the gradient — /root/reference/pytorch/model/heads.py:145-246) compiled for the HOST and run with wave semantics
(tests/host_emul/wave), through their C entry points, against the oracle's restatement (oracle/cbl_oracle.py, itself pinned by the reference's own heads.py:
tests/golden/cbl_pytorch.npz): point mask bit for bit, loss and feature gradient within the kernels' 1e-4 contract.  Every lanes-per-row width of the kernel
(d = 4 ... 64) and neighbour counts on both sides of a wave; the gradient through the scatter entry (no transposed table needed)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import cbl_oracle as C
from tests import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "contrastboundary_amd", "csrc")
EMUL = os.path.join(HERE, "host_emul", "wave")
GEN = os.path.join(HERE, "host_emul", "host_tu.py")
TU = os.path.join(ROOT, "oracle", "_build", "cbl_host.cpp")
SO = os.path.join(ROOT, "oracle", "_build", "libcbl_host.so")


@pytest.fixture(scope="module")
def host():
    srcs = [os.path.join(CSRC, "cbl.hip"), os.path.join(CSRC, "cbl_pairs.hip")]
    deps = srcs + [GEN, os.path.join(CSRC, "wave_ops.h"), os.path.join(CSRC, "cbl_common.h"), os.path.join(EMUL, "amdgcn.h"), os.path.join(EMUL, "hip", "hip_runtime.h")]
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([sys.executable, GEN, TU] + srcs)
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, TU, "-o", SO])
    return ctypes.CDLL(SO)


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def aligned(shape, dtype=np.float32):
    """a zero array whose data pointer is 16-byte aligned (the entries check it: float4 rows)"""
    n = int(np.prod(shape))
    raw = np.zeros(n * np.dtype(dtype).itemsize + 16, np.uint8)
    off = (-raw.ctypes.data) % 16
    return raw[off:off + n * np.dtype(dtype).itemsize].view(dtype).reshape(shape)


def scene(n, nsample, d, seed):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    lab = ((np.floor(xyz[:, 0] * 4) + 4 * np.floor(xyz[:, 1] * 3)).astype(np.int64) % 13).astype(np.int32)     # blocky labels: boundaries between them
    feat = aligned((n, d)); feat[:] = rng.normal(size=(n, d)) * 0.5
    off = np.int32([n // 3, n])
    idx, _ = O.knnquery(nsample, xyz, xyz, off, off)
    return feat, lab, np.ascontiguousarray(idx, np.int32)


@pytest.mark.parametrize("nsample,d,nce", [(8, 16, 0), (17, 32, 0), (33, 4, 0), (36, 32, 0), (40, 64, 0), (65, 8, 0), (17, 32, 1)])
def test_point_contrast(host, nsample, d, nce):
    n, T, weight = 700, 0.7, 0.1
    feat, lab, idx = scene(n, nsample, d, seed=nsample + d)
    per_point, mask, stats, loss = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(2, np.float32), np.zeros(1, np.float32)
    coef, own, grad = np.zeros((n, nsample), np.float32), aligned((n, d)), aligned((n, d))
    flags = 4 if nce else 0
    rc = host.cbl_contrast_pairs_forward(n, 0x7fffffff, flags, nsample, d, P(feat), P(lab), 0, ctypes.c_float(0.0), P(idx), None, ctypes.c_float(T), ctypes.c_float(weight),
                                         P(per_point), P(mask), P(stats), P(loss), P(coef), P(own), None)
    assert rc == 0
    one = np.ones(1, np.float32)
    rc = host.cbl_contrast_pairs_backward_atomic(n, n, nsample, d, P(feat), P(coef), P(own), P(idx), P(stats), P(one), ctypes.c_float(weight), P(grad), None)
    assert rc == 0
    rloss, rgrad, rmask = C.point_contrast(np.array(feat), np.eye(13, dtype=np.float32)[lab], idx, temperature=T, weight=weight, contrast="nce" if nce else "softnn")
    if not nce:
        np.testing.assert_array_equal(mask.astype(bool), rmask)
    assert rmask.any()
    np.testing.assert_allclose(float(loss[0]), float(rloss), rtol=1e-4)
    np.testing.assert_allclose(np.array(grad), rgrad, rtol=1e-4, atol=1e-4 * np.abs(rgrad).max())


def test_gradient_stays_finite_when_every_valid_neighbour_is_far_behind_a_masked_one(host):
    """TF head (head.py:752): the max-shift runs over every column, masked or not.  A centre with an IGNORED coincident neighbour (distance 1e-6, holds the maximum)
    and its valid neighbours 20 away at T = 0.3 has exponentials of 1e-29: P / A is an ordinary number, but A * A underflows in fp32 — the coefficients were
    inf / NaN until they were written as two quotients by A (cbl_pairs.hip), where the reference's autodiff (x / y / y) and the oracle stay finite."""
    n, nsample, d, T, weight = 64, 5, 16, 0.3, 0.1
    rng = np.random.default_rng(0)
    feat = aligned((n, d)); feat[:] = 0
    dirs = rng.normal(size=(48, d)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    feat[16:] = (dirs * 20.0).astype(np.float32)
    lab = np.full(n, -1, np.int32); lab[0] = 0; lab[16:] = np.arange(48) % 2
    idx = np.zeros((n, nsample), np.int32)
    for i in range(n):
        idx[i] = [i, 1, 16 + (i * 3) % 48, 16 + (i * 3 + 1) % 48, 16 + (i * 3 + 2) % 48]       # column 1: the ignored point at the origin
    per_point, mask, stats, loss = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(2, np.float32), np.zeros(1, np.float32)
    coef, own, grad = np.zeros((n, nsample), np.float32), aligned((n, d)), aligned((n, d))
    rc = host.cbl_contrast_pairs_forward(n, n, 1, nsample, d, P(feat), P(lab), 0, ctypes.c_float(0.0), P(idx), None, ctypes.c_float(T), ctypes.c_float(weight),
                                         P(per_point), P(mask), P(stats), P(loss), P(coef), P(own), None)
    assert rc == 0
    one = np.ones(1, np.float32)
    assert host.cbl_contrast_pairs_backward_atomic(n, n, nsample, d, P(feat), P(coef), P(own), P(idx), P(stats), P(one), ctypes.c_float(weight), P(grad), None) == 0
    rloss, rgrad, rmask = C.tf_contrast(np.array(feat), lab, idx, temperature=T, weight=weight)
    assert rmask[0] and rmask.sum() > 10
    np.testing.assert_array_equal(mask.astype(bool), rmask)
    assert np.isfinite(coef).all() and np.isfinite(np.array(grad)).all()
    np.testing.assert_allclose(float(loss[0]), float(rloss), rtol=1e-4)
    np.testing.assert_allclose(np.array(grad), rgrad, rtol=1e-4, atol=1e-4 * np.abs(rgrad).max())
    # the scatter kernels of cbl.hip hold the same expression
    g2, st2, l2, pp2, m2 = aligned((n, d)), np.zeros(2, np.float32), np.zeros(1, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
    g2[:] = 0
    assert host.cbl_tf_contrast_forward(n, n, nsample, d, P(feat), P(lab), P(idx), ctypes.c_float(T), ctypes.c_float(weight), P(pp2), P(m2), P(st2), P(l2), None) == 0
    assert host.cbl_tf_contrast_backward(n, n, nsample, d, P(feat), P(lab), P(idx), ctypes.c_float(T), ctypes.c_float(weight), P(st2), P(one), P(g2), None) == 0
    assert np.isfinite(np.array(g2)).all()
    np.testing.assert_allclose(np.array(g2), rgrad, rtol=1e-4, atol=1e-4 * np.abs(rgrad).max())


def test_negatives_far_below_the_positives_keep_their_sum(host):
    """margin 'S' (head.py:759-760) divides by the negatives' sum alone: with positives at distance 1 and negatives at 8 (T = 0.4) it is 2.5e-8 of the positives',
    below the resolution of A = P + N in fp32 — the kernel sums it separately, as the reference does (A - P was 0, and the term -log(P / 1e-12))."""
    n, nsample, d, T, weight = 96, 9, 8, 0.4, 0.1
    rng = np.random.default_rng(1)
    lab = (np.arange(n) % 2).astype(np.int32)
    feat = aligned((n, d)); feat[:] = rng.normal(size=(n, d)) * 0.05
    feat[:, 0] += np.where(lab == 0, 0.0, 8.0)                        # the two classes 8 apart, points of a class within ~0.2
    idx = np.zeros((n, nsample), np.int32)
    for i in range(n):
        same = [(i + 2 * k) % n for k in range(1, 7)]
        other = [(i + 1) % n, (i + 3) % n]
        idx[i] = [i] + same + other
    per_point, mask, stats, loss = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(2, np.float32), np.zeros(1, np.float32)
    coef, own = np.zeros((n, nsample), np.float32), aligned((n, d))
    rc = host.cbl_contrast_pairs_forward(n, n, 1 | 8, nsample, d, P(feat), P(lab), 0, ctypes.c_float(0.0), P(idx), None, ctypes.c_float(T), ctypes.c_float(weight),
                                         P(per_point), P(mask), P(stats), P(loss), P(coef), P(own), None)
    assert rc == 0
    rloss, _, rmask = C.tf_contrast(np.array(feat), lab, idx, temperature=T, weight=weight, separate=True)
    assert rmask.all()
    np.testing.assert_allclose(float(loss[0]), float(rloss), rtol=1e-4)
    assert float(per_point.max()) < -10                               # -log(P / N) with N ~ 1e-8 P, nowhere near the clamp's -log(P / 1e-12) = -27.6
    assert float(per_point.min()) > -25
