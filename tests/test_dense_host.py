"""CPU: the product's BatchNorm passes (contrastboundary_amd/csrc/bn_rows.hip: statistics + transform + ReLU, with the residual tail of a block,
/root/reference/pytorch/model/blocks.py:126-134) and the criterion's cross entropy (cross_entropy.hip, pointtransformer_seg.py:20-22) compiled for the HOST and run
with wave semantics (tests/host_emul/wave), called through their C entry points and held against torch in float64: outputs, every gradient, running statistics.
Covers the one-kernel path (rows <= 4096), the two-pass path, both lane widths, ignored labels — without a GPU; the `-m gpu` tests hold the device build."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "contrastboundary_amd", "csrc")
EMUL = os.path.join(HERE, "host_emul", "wave")
SRC = os.path.join(HERE, "host_emul", "dense_host.cpp")
SO = os.path.join(ROOT, "oracle", "_build", "libdense_host.so")


@pytest.fixture(scope="module")
def host():
    deps = [SRC, os.path.join(CSRC, "bn_rows.hip"), os.path.join(CSRC, "cross_entropy.hip"), os.path.join(CSRC, "wave_ops.h"), os.path.join(CSRC, "cbl_common.h"),
            os.path.join(EMUL, "amdgcn.h"), os.path.join(EMUL, "hip", "hip_runtime.h")]
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, SRC, "-o", SO])
    L = ctypes.CDLL(SO)
    L.cbl_bn_rows_workspace_bytes.restype = ctypes.c_size_t
    L.cbl_cross_entropy_workspace_bytes.restype = ctypes.c_size_t
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def close(a, ref, tol):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - ref).max()) <= tol * max(float(np.abs(ref).max()), 1e-30)


@pytest.mark.parametrize("rows,C,relu,residual", [(300, 8, 1, True), (4096, 4, 1, False), (100, 16, 0, True),           # one kernel per direction
                                                  (5000, 8, 1, True), (4500, 6, 1, True), (6000, 12, 0, False)])      # two passes; C = 6: one channel per lane
def test_batch_norm_rows_with_residual(host, rows, C, relu, residual):
    bn_case(host, rows, C, relu, residual)


def bn_case(host, rows, C, relu, residual):
    rng = np.random.default_rng(rows + C)
    f = lambda *s: rng.normal(size=s).astype(np.float32)
    x, res, gy = f(rows, C) * 1.7 + 0.3, (f(rows, C) if residual else None), f(rows, C)
    w, b = rng.uniform(0.5, 1.5, C).astype(np.float32), f(C) * 0.5
    rm, rv = f(C), rng.uniform(0.5, 2.0, C).astype(np.float32)
    rm0, rv0 = rm.copy(), rv.copy()
    nbt = np.zeros(1, np.int64)
    mean, invstd, y = np.zeros(C, np.float32), np.zeros(C, np.float32), np.zeros((rows, C), np.float32)
    nbytes = host.cbl_bn_rows_workspace_bytes(ctypes.c_longlong(rows), C)
    ws = np.zeros(nbytes + 64, np.uint8)
    eps, mom = 1e-5, 0.1
    rc = host.cbl_bn_rows_forward_residual(ctypes.c_longlong(rows), C, P(x), P(res), P(w), P(b), ctypes.c_float(eps), ctypes.c_float(mom), P(rm), P(rv), P(nbt), relu,
                                           P(mean), P(invstd), P(y), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    gx, gres = np.zeros((rows, C), np.float32), (np.zeros((rows, C), np.float32) if residual else None)
    gw, gb = np.zeros(C, np.float32), np.zeros(C, np.float32)
    rc = host.cbl_bn_rows_backward_residual(ctypes.c_longlong(rows), C, P(x), P(res), P(gy), P(w), P(b), P(mean), P(invstd), relu, P(gx), P(gres), P(gw), P(gb),
                                            P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    # torch, float64
    bn = torch.nn.BatchNorm1d(C, eps=eps, momentum=mom).double().train()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(w)); bn.bias.copy_(torch.from_numpy(b)); bn.running_mean.copy_(torch.from_numpy(rm0)); bn.running_var.copy_(torch.from_numpy(rv0))
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    r64 = torch.from_numpy(res).double().requires_grad_(True) if residual else None
    ref = bn(x64) + (r64 if residual else 0.0)
    if relu:
        ref = torch.relu(ref)
    ref.backward(torch.from_numpy(gy).double())
    assert close(y, ref.detach().numpy(), 2e-5)
    l2 = lambda a, r: float(np.linalg.norm(np.asarray(a, np.float64) - r) / max(np.linalg.norm(r), 1e-30))      # a ReLU decision may flip where |y| ~ 1e-7
    assert l2(gx, x64.grad.numpy()) < 1e-4 and l2(gw, bn.weight.grad.numpy()) < 1e-4 and l2(gb, bn.bias.grad.numpy()) < 1e-4
    if residual:
        assert l2(gres, r64.grad.numpy()) < 1e-4
    assert close(rm, bn.running_mean.numpy(), 1e-5) and close(rv, bn.running_var.numpy(), 1e-5) and int(nbt[0]) == 1


@pytest.mark.parametrize("n,k,ignored", [(3000, 13, 0.1), (257, 2, 0.5), (1000, 64, 0.0)])
def test_cross_entropy(host, n, k, ignored):
    xe_case(host, n, k, ignored)


def xe_case(host, n, k, ignored):
    rng = np.random.default_rng(n + k)
    z = (rng.normal(size=(n, k)) * 3).astype(np.float32)
    t = rng.integers(0, k, n).astype(np.int64)
    t[rng.uniform(size=n) < ignored] = 255
    loss, stats, g, up = np.zeros(1, np.float32), np.zeros(2, np.float32), np.zeros((n, k), np.float32), np.array([1.7], np.float32)
    nbytes = host.cbl_cross_entropy_workspace_bytes(ctypes.c_longlong(n))
    ws = np.zeros(nbytes + 64, np.uint8)
    assert host.cbl_cross_entropy_forward(ctypes.c_longlong(n), k, P(z), P(t), ctypes.c_longlong(255), P(loss), P(stats), P(ws), ctypes.c_size_t(nbytes), None) == 0
    assert host.cbl_cross_entropy_backward(ctypes.c_longlong(n), k, P(z), P(t), ctypes.c_longlong(255), P(stats), P(up), P(g), None) == 0
    z64 = torch.from_numpy(z).double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(z64, torch.from_numpy(t), ignore_index=255)
    (ref * 1.7).backward()
    assert abs(float(loss[0]) - float(ref.detach())) <= 2e-6 * abs(float(ref.detach()))
    assert int(stats[1]) == int((t != 255).sum())
    assert close(g, z64.grad.numpy(), 2e-6)
    assert not g[t == 255].any()


def test_cross_entropy_label_outside_the_classes_is_loud(host):
    """a label outside [0, k) that is not ignore_index (255 against ignore_label 0: a mapping bug) — nn.CrossEntropyLoss device-asserts; the fused kernel
    returns a NaN loss instead of quietly training on fewer points, and a zero gradient row for that point"""
    n, k = 500, 13
    rng = np.random.default_rng(3)
    z = rng.normal(size=(n, k)).astype(np.float32)
    t = rng.integers(0, k, n).astype(np.int64)
    t[7] = 255
    loss, stats, g, up = np.zeros(1, np.float32), np.zeros(2, np.float32), np.ones((n, k), np.float32), np.array([1.0], np.float32)
    nbytes = host.cbl_cross_entropy_workspace_bytes(ctypes.c_longlong(n))
    ws = np.zeros(nbytes + 64, np.uint8)
    assert host.cbl_cross_entropy_forward(ctypes.c_longlong(n), k, P(z), P(t), ctypes.c_longlong(-100), P(loss), P(stats), P(ws), ctypes.c_size_t(nbytes), None) == 0
    assert np.isnan(loss[0])
    assert host.cbl_cross_entropy_backward(ctypes.c_longlong(n), k, P(z), P(t), ctypes.c_longlong(-100), P(stats), P(up), P(g), None) == 0
    assert not g[7].any()


@pytest.mark.skipif(not os.environ.get("CBL_HOST_EMUL_FULL"), reason="a second (sanitizer) build of the host library: set CBL_HOST_EMUL_FULL=1")
def test_kernels_under_address_sanitizer(tmp_path):
    """the same host build with -fsanitize=address in a subprocess: operands are numpy buffers of exactly their logical sizes, `__shared__` arrays static arrays"""
    import sys
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan beside gcc")
    so = os.path.join(ROOT, "oracle", "_build", "libdense_host_asan.so")
    deps = [SRC, os.path.join(CSRC, "bn_rows.hip"), os.path.join(CSRC, "cross_entropy.hip"), os.path.join(CSRC, "wave_ops.h"), os.path.join(CSRC, "cbl_common.h"),
            os.path.join(EMUL, "amdgcn.h"), os.path.join(EMUL, "hip", "hip_runtime.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, SRC, "-o", so])
    script = tmp_path / "run.py"
    script.write_text(
        "import ctypes, sys\n"
        "sys.path.insert(0, %r)\n"
        "import tests.test_dense_host as T\n"
        "L = ctypes.CDLL(%r)\n"
        "L.cbl_bn_rows_workspace_bytes.restype = ctypes.c_size_t; L.cbl_cross_entropy_workspace_bytes.restype = ctypes.c_size_t\n"
        "for rows, C, relu, res in ((301, 8, 1, True), (4097, 4, 1, True), (4500, 6, 1, True)):\n"
        "    T.bn_case(L, rows, C, relu, res)\n"
        "T.xe_case(L, 1001, 13, 0.2)\n"
        "print('ASAN_RUN_DONE')\n" % (ROOT, so))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "ASAN_RUN_DONE" in r.stdout, (r.returncode, r.stderr[-2000:])
