"""CPU: the product's Point-Transformer layer kernels (contrastboundary_amd/csrc/pt_layer.hip, row a4) compiled for the HOST and run with wave
semantics (tests/host_emul/wave: every thread a fibre, MFMA / DPP / barriers as rendezvous) on small shapes, against the layer's formula
(/root/reference/pytorch/model/blocks.py:31-44, train-mode BatchNorms) written with torch ops in float64 and differentiated by autograd.
This checks the tile layouts, the index arithmetic and every gradient formula of the kernels without a GPU; the `-m gpu` tests check the device build."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(ROOT, "contrastboundary_amd", "csrc", "pt_layer.hip")
EMUL = os.path.join(HERE, "host_emul", "wave")
SO = os.path.join(ROOT, "oracle", "_build", "libpt_layer_host.so")


@pytest.fixture(scope="module")
def host():
    deps = [SRC, os.path.join(EMUL, "pt_wave.h"), os.path.join(EMUL, "hip", "hip_runtime.h"), os.path.join(ROOT, "contrastboundary_amd", "csrc", "cbl_common.h")]
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "contrastboundary_amd", "csrc"), SRC, "-o", SO])
    L = ctypes.CDLL(SO)
    L.cbl_pt_layer_workspace_bytes.restype = ctypes.c_size_t
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def knn(xyz, K):
    d = ((xyz[:, None, :] - xyz[None, :, :]) ** 2).sum(-1)
    return np.argsort(d, axis=1, kind="stable")[:, :K].astype(np.int32)


def make(n, K, C, seed):
    rng = np.random.default_rng(seed)
    G = C // 8
    f = lambda *s: rng.normal(size=s).astype(np.float32)
    t = dict(xyz=rng.uniform(0, 1, (n, 3)).astype(np.float32), x_q=f(n, C), x_k=f(n, C), x_v=f(n, C),
             Wp=f(3, 3) * 2, bp=f(3) * 0.1, gamma_p=rng.uniform(0.5, 1.5, 3).astype(np.float32), beta_p=f(3) * 0.2,
             W3C=f(C, 3) * 0.5, b3C=f(C) * 0.1, gamma_c=rng.uniform(0.5, 1.5, C).astype(np.float32), beta_c=f(C) * 0.2,
             Wa=f(G, C) * 0.3, ba=f(G) * 0.1, gamma_g=rng.uniform(0.5, 1.5, G).astype(np.float32), beta_g=f(G) * 0.2,
             Wb=f(G, G) * 0.6, bb=f(G) * 0.1, g_out=f(n, C))
    t["idx"] = knn(t["xyz"], K)
    return t


PARAMS = ["Wp", "bp", "gamma_p", "beta_p", "W3C", "b3C", "gamma_c", "beta_c", "Wa", "ba", "gamma_g", "beta_g", "Wb", "bb"]
EPS = 1e-5


def reference(t, K, C):
    """blocks.py:31-44 behind the q/k/v Linear layers, float64, autograd"""
    G = C // 8
    T = {k: torch.tensor(v, dtype=torch.float64, requires_grad=(k in PARAMS or k in ("x_q", "x_k", "x_v"))) for k, v in t.items() if k != "idx"}
    idx = torch.tensor(t["idx"].astype(np.int64))
    n = idx.shape[0]

    def bn(x, g, b):                                                  # train-mode BatchNorm1d over all rows
        m = x.mean(0); v = x.var(0, unbiased=False)
        return (x - m) / torch.sqrt(v + EPS) * g + b, m, v

    p_r = T["xyz"][idx] - T["xyz"][:, None, :]
    p0 = p_r @ T["Wp"].T + T["bp"]
    y, mp, vp = bn(p0.reshape(n * K, 3), T["gamma_p"], T["beta_p"])
    p1 = torch.relu(y).reshape(n, K, 3)
    pe = p1 @ T["W3C"].T + T["b3C"]
    w = T["x_k"][idx] - T["x_q"][:, None, :] + pe
    y, mc, vc = bn(w.reshape(n * K, C), T["gamma_c"], T["beta_c"])
    w2 = torch.relu(y) @ T["Wa"].T + T["ba"]
    y, mg, vg = bn(w2, T["gamma_g"], T["beta_g"])
    logits = (torch.relu(y) @ T["Wb"].T + T["bb"]).reshape(n, K, G)
    a = torch.softmax(logits, dim=1)
    out = ((T["x_v"][idx] + pe).view(n, K, 8, G) * a.unsqueeze(2)).sum(1).view(n, C)
    (out * T["g_out"]).sum().backward()
    grads = {k: T[k].grad.numpy() for k in PARAMS + ["x_q", "x_k", "x_v"]}
    stats = dict(mean=[mp, mc, mg], var=[vp, vc, vg])
    return out.detach().numpy(), grads, dict(p1=p1.detach().numpy(), w2=w2.detach().reshape(n, K, G).numpy(), a=a.detach().numpy()), stats


def transpose_table(idx, order):
    n, K = idx.shape
    flat = idx.reshape(-1)
    src = np.argsort(flat, kind="stable").astype(np.int32)            # pairs grouped by target, ascending pair id inside a target
    counts = np.bincount(flat, minlength=n)
    first = np.concatenate([[0], np.cumsum(counts)])
    ranks = np.arange(n) if order is None else order
    inv_start = np.zeros(n + 1, np.int32); inv_src = np.zeros(n * K, np.int32)
    o = 0
    for tr, j in enumerate(ranks):
        inv_start[tr] = o
        seg = src[first[j]:first[j + 1]]
        inv_src[o:o + len(seg)] = seg
        o += len(seg)
    inv_start[n] = o
    return inv_start, inv_src


def run_host(L, t, K, C, order):
    n = t["xyz"].shape[0]
    G = C // 8
    z = lambda *s: np.zeros(s, np.float32)
    f32 = ctypes.c_float
    buf = dict(p_r=z(n, K, 3), p0=z(n, K, 3), p1=z(n, K, 3), w2=z(n, K, G), a=z(n, K, G), out=z(n, C), consts=z(L.cbl_pt_layer_consts_floats()))
    ws = np.zeros(L.cbl_pt_layer_workspace_bytes(n, K, C) // 4 + 16, np.float32)
    rm = [z(3), z(C), z(G)]; rv = [np.ones(3, np.float32), np.ones(C, np.float32), np.ones(G, np.float32)]; nb = [np.zeros(1, np.int64) for _ in range(3)]
    arr3 = lambda xs: (ctypes.c_void_p * 3)(*[x.ctypes.data for x in xs])
    eps3 = (f32 * 3)(EPS, EPS, EPS); mom3 = (f32 * 3)(0.1, 0.1, 0.1)
    rc = L.cbl_pt_layer_forward(n, K, C, P(t["xyz"]), P(t["x_q"]), P(t["x_k"]), P(t["x_v"]), P(t["idx"]), P(order), *[P(t[k]) for k in PARAMS], eps3, mom3,
                                arr3(rm), arr3(rv), arr3(nb), P(buf["p_r"]), P(buf["p0"]), P(buf["p1"]), P(buf["w2"]), P(buf["a"]), P(buf["out"]), P(buf["consts"]),
                                P(ws), ctypes.c_size_t(ws.nbytes), None)
    assert rc == 0, rc
    inv_start, inv_src = transpose_table(t["idx"], order)
    g = {k: np.zeros_like(t[k]) for k in PARAMS + ["x_q", "x_k", "x_v"]}
    rc = L.cbl_pt_layer_backward(n, K, C, P(t["x_q"]), P(t["x_k"]), P(t["x_v"]), P(t["idx"]), P(order), P(inv_start), P(inv_src), P(t["gamma_p"]), P(t["W3C"]),
                                 P(t["b3C"]), P(t["gamma_c"]), P(t["Wa"]), P(t["gamma_g"]), P(t["Wb"]), P(buf["p_r"]), P(buf["p0"]), P(buf["p1"]), P(buf["w2"]),
                                 P(buf["a"]), P(buf["consts"]), P(t["g_out"]), P(g["x_q"]), P(g["x_k"]), P(g["x_v"]), *[P(g[k]) for k in PARAMS],
                                 P(ws), ctypes.c_size_t(ws.nbytes), None)
    assert rc == 0, rc
    return buf, g, dict(rm=rm, rv=rv, nb=nb)


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("n,K,C,ordered", [(40, 16, 64, True), (37, 8, 32, True), (24, 16, 32, False), (29, 8, 64, False), (16, 16, 32, True), (17, 8, 64, True)])
def test_layer_kernels_on_the_host_against_autograd(host, n, K, C, ordered):
    t = make(n, K, C, seed=n + C)
    order = np.random.default_rng(1).permutation(n).astype(np.int32) if ordered else None
    out, grads, mid, stats = reference(t, K, C)
    buf, g, run = run_host(host, t, K, C, order)
    assert rel(buf["p1"], mid["p1"]) < 1e-5
    assert rel(buf["w2"], mid["w2"]) < 1e-5
    assert rel(buf["a"], mid["a"]) < 1e-5
    assert rel(buf["out"], out) < 1e-5
    # running statistics as torch's train-mode BatchNorm1d: momentum 0.1, unbiased variance
    rows = n * K
    for q in range(3):
        np.testing.assert_allclose(run["rm"][q], 0.1 * stats["mean"][q].detach().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(run["rv"][q], 0.9 + 0.1 * stats["var"][q].detach().numpy() * rows / (rows - 1), rtol=1e-4, atol=1e-6)
        assert run["nb"][q][0] == 1
    gmax = max(float(np.abs(v).max()) for v in grads.values())
    for k in ["x_v", "x_q", "x_k"] + PARAMS:
        # biases in front of a BatchNorm have a zero true gradient: both sides return rounding noise (absolute bound relative to the largest gradient)
        assert rel(g[k], grads[k]) < 2e-4 or float(np.abs(g[k] - grads[k]).max()) < 1e-5 * gmax, (k, rel(g[k], grads[k]))


@pytest.mark.parametrize("n,K,C", [(33, 16, 64), (26, 8, 32)])
def test_evaluation_mode_on_the_host_uses_the_running_statistics(host, n, K, C):
    """cbl_pt_layer_forward_eval (nn.Module.eval(): the three BatchNorms normalise with their running statistics, no statistics passes, no buffer touched)
    against the layer's formula in float64 with the same running statistics"""
    t = make(n, K, C, seed=3 * n + C)
    G = C // 8
    rng = np.random.default_rng(n)
    rm = [rng.normal(size=d).astype(np.float32) * 0.3 for d in (3, C, G)]
    rv = [rng.uniform(0.5, 2.0, size=d).astype(np.float32) for d in (3, C, G)]
    z = lambda *s: np.zeros(s, np.float32)
    buf = dict(p_r=z(n, K, 3), p0=z(n, K, 3), p1=z(n, K, 3), w2=z(n, K, G), a=z(n, K, G), out=z(n, C), consts=z(host.cbl_pt_layer_consts_floats()))
    ws = np.zeros(host.cbl_pt_layer_workspace_bytes(n, K, C) // 4 + 16, np.float32)
    arr3 = lambda xs: (ctypes.c_void_p * 3)(*[x.ctypes.data for x in xs])
    eps3 = (ctypes.c_float * 3)(EPS, EPS, EPS)
    keep = [x.copy() for x in rm + rv]
    rc = host.cbl_pt_layer_forward_eval(n, K, C, P(t["xyz"]), P(t["x_q"]), P(t["x_k"]), P(t["x_v"]), P(t["idx"]), None, *[P(t[k]) for k in PARAMS], eps3,
                                        arr3(rm), arr3(rv), P(buf["p_r"]), P(buf["p0"]), P(buf["p1"]), P(buf["w2"]), P(buf["a"]), P(buf["out"]), P(buf["consts"]),
                                        P(ws), ctypes.c_size_t(ws.nbytes), None)
    assert rc == 0, rc
    for a, b in zip(rm + rv, keep):
        assert np.array_equal(a, b)                                           # evaluation mode leaves the running statistics alone
    T = {k: torch.tensor(v, dtype=torch.float64) for k, v in t.items() if k != "idx"}
    idx = torch.tensor(t["idx"].astype(np.int64))
    bn = lambda x, q, g, b: (x - torch.tensor(rm[q], dtype=torch.float64)) / torch.sqrt(torch.tensor(rv[q], dtype=torch.float64) + EPS) * g + b
    p_r = T["xyz"][idx] - T["xyz"][:, None, :]
    p1 = torch.relu(bn(p_r @ T["Wp"].T + T["bp"], 0, T["gamma_p"], T["beta_p"]))
    pe = p1 @ T["W3C"].T + T["b3C"]
    w = T["x_k"][idx] - T["x_q"][:, None, :] + pe
    w2 = torch.relu(bn(w, 1, T["gamma_c"], T["beta_c"])) @ T["Wa"].T + T["ba"]
    logits = torch.relu(bn(w2, 2, T["gamma_g"], T["beta_g"])) @ T["Wb"].T + T["bb"]
    a = torch.softmax(logits, dim=1)
    out = ((T["x_v"][idx] + pe).view(n, K, 8, G) * a.unsqueeze(2)).sum(1).view(n, C)
    assert rel(buf["a"], a.numpy()) < 1e-5
    assert rel(buf["out"], out.numpy()) < 1e-5


def test_targets_with_long_pair_lists_on_the_host(host):
    """a neighbour table in which three points are almost everybody's neighbours: their lists in the transposed table hold hundreds of pairs (the
    gather pass over the table walks them in batches), every other point's list is short or empty"""
    n, K, C = 48, 16, 32
    t = make(n, K, C, seed=77)
    rng = np.random.default_rng(5)
    idx = t["idx"].copy()
    idx[:, 1:] = rng.integers(0, 3, size=(n, K - 1))
    idx[5:9, 1:4] = rng.integers(3, n, size=(4, 3))
    t["idx"] = idx.astype(np.int32)
    assert np.bincount(t["idx"].reshape(-1), minlength=n).max() > 64
    out, grads, mid, stats = reference(t, K, C)
    buf, g, run = run_host(host, t, K, C, np.random.default_rng(2).permutation(n).astype(np.int32))
    assert rel(buf["out"], out) < 1e-5
    gmax = max(float(np.abs(v).max()) for v in grads.values())
    for k in ["x_v", "x_q", "x_k"] + PARAMS:
        assert rel(g[k], grads[k]) < 2e-4 or float(np.abs(g[k] - grads[k]).max()) < 1e-5 * gmax, (k, rel(g[k], grads[k]))


def test_kernels_under_address_sanitizer(tmp_path):
    """The same host build with -fsanitize=address, one forward + backward per template shape in a subprocess (libasan must be the first library of the
    process): `__shared__` arrays are static arrays there, so an LDS index past an array's end — which the device answers with another workgroup's LDS and
    the plain host build with whatever static follows — is a reported global-buffer-overflow.  (Round 5: the apply pass's operand table was sized 4 C floats
    for 16 C; every host test passed, the device returned a wrong d p1.)"""
    import sys
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan beside gcc")
    so = os.path.join(ROOT, "oracle", "_build", "libpt_layer_host_asan.so")
    deps = [SRC, os.path.join(EMUL, "pt_wave.h"), os.path.join(EMUL, "hip", "hip_runtime.h"), os.path.join(ROOT, "contrastboundary_amd", "csrc", "cbl_common.h")]
    os.makedirs(os.path.dirname(so), exist_ok=True)
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "contrastboundary_amd", "csrc"), SRC, "-o", so])
    script = tmp_path / "run.py"
    script.write_text(
        "import ctypes, sys\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import tests.test_pt_layer_host as T\n"
        "L = ctypes.CDLL(%r)\n"
        "L.cbl_pt_layer_workspace_bytes.restype = ctypes.c_size_t\n"
        "for n, K, C in ((40, 16, 64), (37, 8, 32), (24, 16, 32), (29, 8, 64)):\n"
        "    t = T.make(n, K, C, seed=n + C)\n"
        "    T.run_host(L, t, K, C, np.random.default_rng(1).permutation(n).astype(np.int32))\n"
        "print('ASAN_RUN_DONE')\n" % (ROOT, so))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "ASAN_RUN_DONE" in r.stdout, (r.returncode, r.stderr[-2000:])
