"""CPU: the product's Linear kernels for many rows and small widths (contrastboundary_amd/csrc/skinny_linear.hip: /root/reference/pytorch/model/blocks.py:23-28,33,62-76)
compiled for the HOST and run with wave semantics (tests/host_emul/wave: v_mfma_f32_16x16x4_f32 as a rendezvous of the wave's fibres), through their C entry points,
against numpy in float64: forward, input gradient, weight and bias gradient — the MFMA walks (widths 16 / 32 / 48 / 64), the ragged walks with operands padded in
registers (c_in = 35, 19, 50, 63: index arithmetic that reads and writes rows of their TRUE stride), and the streaming kernels for everything else."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "contrastboundary_amd", "csrc")
EMUL = os.path.join(HERE, "host_emul", "wave")
GEN = os.path.join(HERE, "host_emul", "host_tu.py")
TU = os.path.join(ROOT, "oracle", "_build", "skinny_linear_host.cpp")
SO = os.path.join(ROOT, "oracle", "_build", "libskinny_linear_host.so")


@pytest.fixture(scope="module")
def host():
    src = os.path.join(CSRC, "skinny_linear.hip")
    deps = [src, GEN, os.path.join(CSRC, "cbl_common.h"), os.path.join(EMUL, "amdgcn.h"), os.path.join(EMUL, "hip", "hip_runtime.h")]
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([sys.executable, GEN, TU, src])
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, TU, "-o", SO])
    L = ctypes.CDLL(SO)
    L.cbl_skinny_linear_workspace_bytes.restype = ctypes.c_size_t
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("rows,cin,cout,bias", [(530, 35, 64, True), (1000, 35, 64, False), (333, 19, 32, True), (400, 50, 16, False), (257, 63, 48, True),   # ragged c_in
                                                (530, 32, 32, True), (300, 64, 16, False), (200, 16, 48, True), (129, 64, 64, True),                         # whole tiles
                                                (700, 7, 5, True), (900, 3, 3, True), (500, 64, 8, False), (300, 6, 32, True)])                             # streaming
def test_linear_forward_and_gradients(host, rows, cin, cout, bias):
    run_case(host, rows, cin, cout, bias)


def run_case(host, rows, cin, cout, bias):
    rng = np.random.default_rng(rows + cin * 7 + cout)
    x = rng.normal(size=(rows, cin)).astype(np.float32)
    w = (rng.normal(size=(cout, cin)) / np.sqrt(cin)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32) if bias else None
    gy = rng.normal(size=(rows, cout)).astype(np.float32)
    y, gx, gw = np.full((rows, cout), np.nan, np.float32), np.full((rows, cin), np.nan, np.float32), np.full((cout, cin), np.nan, np.float32)
    gb = np.full(cout, np.nan, np.float32) if bias else None
    R = ctypes.c_longlong(rows)
    assert host.cbl_skinny_linear_forward(R, cin, cout, P(x), P(w), P(b), P(y), None) == 0
    assert host.cbl_skinny_linear_backward_input(R, cin, cout, P(gy), P(w), P(gx), None) == 0
    nbytes = host.cbl_skinny_linear_workspace_bytes(cin, cout)
    ws = np.zeros(nbytes + 64, np.uint8)
    assert host.cbl_skinny_linear_backward_weight(R, cin, cout, P(x), P(gy), P(gw), P(gb), P(ws), ctypes.c_size_t(nbytes), None) == 0
    x64, w64, g64 = x.astype(np.float64), w.astype(np.float64), gy.astype(np.float64)
    close = lambda a, ref, tol: float(np.abs(a.astype(np.float64) - ref).max()) <= tol * max(float(np.abs(ref).max()), 1e-30)
    assert close(y, x64 @ w64.T + (b.astype(np.float64) if bias else 0.0), 1e-5)
    assert close(gx, g64 @ w64, 1e-5)
    assert close(gw, g64.T @ x64, 2e-5)
    if bias:
        assert close(gb, g64.sum(0), 2e-5)


@pytest.mark.skipif(not os.environ.get("CBL_HOST_EMUL_FULL"), reason="a second (sanitizer) build of the host library: set CBL_HOST_EMUL_FULL=1")
def test_kernels_under_address_sanitizer(tmp_path):
    """The same host build with -fsanitize=address in a subprocess (libasan first): the operands are numpy buffers of exactly rows x c_in / rows x c_out floats
    with red zones behind them, so a ragged walk that reads or writes past a row of the LAST tile — clamped loads, masked stores — is a reported heap overflow."""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan beside gcc")
    src = os.path.join(CSRC, "skinny_linear.hip")
    tu = os.path.join(ROOT, "oracle", "_build", "skinny_linear_host_asan.cpp")
    so = os.path.join(ROOT, "oracle", "_build", "libskinny_linear_host_asan.so")
    deps = [src, GEN, os.path.join(CSRC, "cbl_common.h"), os.path.join(EMUL, "amdgcn.h"), os.path.join(EMUL, "hip", "hip_runtime.h")]
    os.makedirs(os.path.dirname(so), exist_ok=True)
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call([sys.executable, GEN, tu, src])
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, tu, "-o", so])
    script = tmp_path / "run.py"
    script.write_text(
        "import ctypes, sys\n"
        "sys.path.insert(0, %r)\n"
        "import tests.test_skinny_linear_host as T\n"
        "L = ctypes.CDLL(%r)\n"
        "L.cbl_skinny_linear_workspace_bytes.restype = ctypes.c_size_t\n"
        "for rows, cin, cout, bias in ((530, 35, 64, True), (333, 19, 32, True), (257, 63, 48, True), (401, 50, 16, False), (129, 64, 64, True), (700, 7, 5, True)):\n"
        "    T.run_case(L, rows, cin, cout, bias)\n"
        "print('ASAN_RUN_DONE')\n" % (ROOT, so))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "ASAN_RUN_DONE" in r.stdout, (r.returncode, r.stderr[-2000:])
