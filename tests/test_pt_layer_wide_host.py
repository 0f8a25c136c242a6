"""CPU: the wide stages of the Point Transformer's attention layer (C = 128 / 256: `cbl_pt_layer_wide_forward / _backward` in contrastboundary_amd/csrc/pt_layer.hip —
native host code issuing the kernels of attention.hip, the p chain of pt_layer.hip and the narrow pw_* kernels; /root/reference/pytorch/model/blocks.py:31-44 at the
three coarse stages) compiled for the HOST and run with wave semantics (tests/host_emul/wave), against the layer's formula in float64 differentiated by autograd
(the reference function of tests/test_pt_layer_host.py): saved tensors, output, running statistics, the gradient of every input and parameter."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.test_pt_layer_host import EPS, PARAMS, make, reference, rel

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "contrastboundary_amd", "csrc")
EMUL = os.path.join(HERE, "host_emul", "wave")
GEN = os.path.join(HERE, "host_emul", "host_tu.py")
BUILD = os.path.join(ROOT, "oracle", "_build")
SO = os.path.join(BUILD, "libpt_layer_wide_host.so")


@pytest.fixture(scope="module")
def host():
    srcs = {"pt_layer": os.path.join(CSRC, "pt_layer.hip"), "attention": os.path.join(CSRC, "attention.hip")}
    deps = list(srcs.values()) + [GEN, os.path.abspath(__file__), os.path.join(CSRC, "cbl_common.h"), os.path.join(EMUL, "pt_wave.h"), os.path.join(EMUL, "amdgcn.h"),
                                  os.path.join(EMUL, "hip", "hip_runtime.h")]
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        objs = []
        for name, src in srcs.items():
            tu, obj = os.path.join(BUILD, name + "_widehost.cpp"), os.path.join(BUILD, name + "_widehost.o")
            # pt_layer.hip WITH its wide-stage section (its older host build stops in front of it)
            subprocess.check_call([sys.executable, GEN] + (["--whole"] if name == "pt_layer" else []) + [tu, src])
            subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-c", "-ffp-contract=off", "-Wno-unknown-pragmas",
                                   "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, tu, "-o", obj])
            objs.append(obj)
        subprocess.check_call(["g++", "-shared"] + objs + ["-o", SO])
    L = ctypes.CDLL(SO)
    L.cbl_pt_layer_wide_workspace_bytes.restype = ctypes.c_size_t
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def aligned(a):
    a = np.ascontiguousarray(a)
    raw = np.zeros(a.nbytes + 16, np.uint8)
    off = (-raw.ctypes.data) % 16
    out = raw[off:off + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


@pytest.mark.parametrize("n,K,C", [(40, 16, 128), (23, 16, 256), (35, 8, 128), (17, 16, 512)])
def test_wide_layer_on_the_host_against_autograd(host, n, K, C):
    t = {k: aligned(v) for k, v in make(n, K, C, seed=n + C).items()}
    G = C // 8
    out, grads, mid, stats = reference(t, K, C)
    z = lambda *s: aligned(np.zeros(s, np.float32))
    f32 = ctypes.c_float
    buf = dict(p_r=z(n, K, 3), p0=z(n, K, 3), p1=z(n, K, 3), w2=z(n, K, G), a=z(n, K, G), out=z(n, C), consts=z(host.cbl_pt_layer_wide_consts_floats()), bnc=z(2 * C))
    nbytes = host.cbl_pt_layer_wide_workspace_bytes(n, K, C)
    ws = aligned(np.zeros(nbytes // 4 + 16, np.float32))
    rm = [z(3), z(C), z(G)]; rv = [aligned(np.ones(d, np.float32)) for d in (3, C, G)]; nb = [np.zeros(1, np.int64) for _ in range(3)]
    arr3 = lambda xs: (ctypes.c_void_p * 3)(*[x.ctypes.data for x in xs])
    eps3 = (f32 * 3)(EPS, EPS, EPS); mom3 = (f32 * 3)(0.1, 0.1, 0.1)
    rc = host.cbl_pt_layer_wide_forward(n, K, C, P(t["xyz"]), P(t["x_q"]), P(t["x_k"]), P(t["x_v"]), P(t["idx"]), *[P(t[k]) for k in PARAMS], eps3, mom3,
                                        arr3(rm), arr3(rv), arr3(nb), P(buf["p_r"]), P(buf["p0"]), P(buf["p1"]), P(buf["w2"]), P(buf["a"]), P(buf["out"]), P(buf["consts"]),
                                        P(buf["bnc"]), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0, rc
    assert rel(buf["p1"], mid["p1"]) < 1e-5 and rel(buf["w2"], mid["w2"]) < 1e-5 and rel(buf["a"], mid["a"]) < 1e-5 and rel(buf["out"], out) < 1e-5
    rows = n * K
    for q in range(3):
        np.testing.assert_allclose(rm[q], 0.1 * stats["mean"][q].detach().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(rv[q], 0.9 + 0.1 * stats["var"][q].detach().numpy() * rows / (rows - 1), rtol=1e-4, atol=1e-6)
        assert nb[q][0] == 1
    g = {k: aligned(np.full(t[k].shape, np.nan, np.float32)) for k in PARAMS}
    g_qkv = aligned(np.full((3, n, C), np.nan, np.float32))            # d x_k and d x_v adjacent: the call zeroes both scatter targets with one fill
    rc = host.cbl_pt_layer_wide_backward(n, K, C, P(t["x_q"]), P(t["x_k"]), P(t["x_v"]), P(t["idx"]), P(t["gamma_p"]), P(t["W3C"]), P(t["b3C"]), P(t["gamma_c"]),
                                         P(t["beta_c"]), P(t["Wa"]), P(t["gamma_g"]), P(t["Wb"]), P(buf["p_r"]), P(buf["p0"]), P(buf["p1"]), P(buf["w2"]), P(buf["a"]),
                                         P(buf["consts"]), P(buf["bnc"]), P(t["g_out"]), P(g_qkv[0]), P(g_qkv[1]), P(g_qkv[2]), *[P(g[k]) for k in PARAMS],
                                         P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0, rc
    got = dict(g, x_q=g_qkv[0], x_k=g_qkv[1], x_v=g_qkv[2])
    gmax = max(float(np.abs(v).max()) for v in grads.values())
    for k in ["x_v", "x_q", "x_k"] + PARAMS:
        assert rel(got[k], grads[k]) < 2e-4 or float(np.abs(got[k] - grads[k]).max()) < 1e-5 * gmax, (k, rel(got[k], grads[k]))
