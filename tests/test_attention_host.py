"""CPU: the C-wide passes of the Point Transformer layer (contrastboundary_amd/csrc/attention.hip: /root/reference/pytorch/model/blocks.py:31-44 without its (n, K, C)
tensors — the relative-position encoding p_r = Linear(3, C)(p1) recomputed on the fly) run from the host build of the whole library (tests/host_emul/full_library.py,
wave semantics) through their C entry points, against the same expressions under torch autograd in float64:
    attn_w2   w2 = Linear(C, G)(ReLU(BN_C(x_k[idx] - x_q + p_r)))              blocks.py:39 and the first half of linear_w (:25-27), train-mode batch statistics
    attn_agg  out = sum_k (x_v[idx] + p_r) * a[..., c % G]                       blocks.py:42-43, also with the softmax over the K neighbours (:41) inside
both with the scatter (float atomics) backward and with the gather over the transposed neighbour table (cbl_neighbor_transpose), and the three per-point
projections of blocks.py:33 as one launch per direction (cbl_triple_linear_*).  C = 32 / 64 (the fused-layer stages) and 128 (a wide stage)."""
import ctypes

import numpy as np
import pytest
import torch

from tests.host_emul import full_library

F = ctypes.c_float


@pytest.fixture(scope="module")
def host():
    L = full_library.load()
    for name in ("cbl_attn_workspace_bytes", "cbl_neighbor_transpose_workspace_bytes", "cbl_triple_linear_workspace_bytes"):
        getattr(L, name).restype = ctypes.c_size_t
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def close(got, ref, tol=2e-4):
    ref = np.asarray(ref, np.float64)
    scale = max(float(np.abs(ref).max()), 1e-30)
    err = float(np.abs(np.asarray(got, np.float64) - ref).max()) / scale
    assert err < tol, err


def scene(n, K, C, seed):
    rng = np.random.default_rng(seed)
    G = C // 8
    idx = rng.integers(0, n, (n, K)).astype(np.int32)
    idx[:, 0] = np.arange(n)
    idx[n // 3] = 7                                                   # a row that lists one point K times: one segment of the transposed table holds them all
    a = dict(x_q=rng.normal(size=(n, C)), x_k=rng.normal(size=(n, C)), x_v=rng.normal(size=(n, C)), p1=np.abs(rng.normal(size=(n, K, 3))),
             W3C=rng.normal(size=(C, 3)) * 0.5, b3C=rng.normal(size=C) * 0.1, gamma=rng.uniform(0.5, 1.5, C), beta=rng.normal(size=C) * 0.1,
             Wa=rng.normal(size=(G, C)) / np.sqrt(C), ba=rng.normal(size=G) * 0.1, logits=rng.normal(size=(n, K, G)),
             g_w2=rng.normal(size=(n, K, G)), g_out=rng.normal(size=(n, C)))
    return idx, {k: np.ascontiguousarray(v, np.float32) for k, v in a.items()}


def transposed(host, idx):
    n, K = idx.shape
    inv_start, inv_src = np.full(n + 1, -1, np.int32), np.full(n * K, -1, np.int32)
    nbytes = host.cbl_neighbor_transpose_workspace_bytes(n, n, K)
    ws = np.zeros(nbytes + 64, np.uint8)
    assert host.cbl_neighbor_transpose(n, n, K, P(idx), None, None, P(inv_start), P(inv_src), P(ws), ctypes.c_size_t(nbytes), None) == 0
    return inv_start, inv_src


@pytest.mark.parametrize("n,K,C", [(300, 16, 32), (260, 8, 64), (150, 16, 128)])
def test_attention_logits_pass(host, n, K, C):
    G, eps, mom = C // 8, 1e-5, 0.1
    idx, a = scene(n, K, C, seed=C + K)
    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in a.items()}
    ti = torch.from_numpy(idx.astype(np.int64))
    p_r = t["p1"] @ t["W3C"].T + t["b3C"]
    pre = t["x_k"][ti] - t["x_q"][:, None, :] + p_r
    flat = pre.reshape(-1, C)
    mean, var = flat.mean(0), flat.var(0, unbiased=False)
    h = torch.relu((pre - mean) / torch.sqrt(var + eps) * t["gamma"] + t["beta"])
    w2 = h @ t["Wa"].T + t["ba"]
    (w2 * t["g_w2"].detach()).sum().backward()
    # forward (training): statistics, running statistics as nn.BatchNorm1d updates them, logits
    run_mean, run_var, count = np.zeros(C, np.float32), np.ones(C, np.float32), np.zeros(1, np.int64)
    save_mean, save_invstd, out = np.full(C, np.nan, np.float32), np.full(C, np.nan, np.float32), np.full((n, K, G), np.nan, np.float32)
    nbytes = host.cbl_attn_workspace_bytes(C, G)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = host.cbl_attn_w2_forward(n, K, C, G, P(a["x_q"]), P(a["x_k"]), P(idx), P(a["p1"]), P(a["W3C"]), P(a["b3C"]), P(a["gamma"]), P(a["beta"]), F(eps), F(mom),
                                  P(run_mean), P(run_var), P(count), 1, P(a["Wa"]), P(a["ba"]), P(save_mean), P(save_invstd), P(out), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    close(out, w2.detach().numpy())
    close(save_mean, mean.detach().numpy()); close(save_invstd, (1.0 / torch.sqrt(var + eps)).detach().numpy())
    close(run_mean, mom * mean.detach().numpy()); close(run_var, 0.9 + mom * flat.var(0, unbiased=True).detach().numpy())
    assert int(count[0]) == 1
    # evaluation mode: the saved arrays are inputs
    ev = np.full((n, K, G), np.nan, np.float32)
    rc = host.cbl_attn_w2_forward(n, K, C, G, P(a["x_q"]), P(a["x_k"]), P(idx), P(a["p1"]), P(a["W3C"]), P(a["b3C"]), P(a["gamma"]), P(a["beta"]), F(eps), F(mom),
                                  None, None, None, 0, P(a["Wa"]), P(a["ba"]), P(save_mean), P(save_invstd), P(ev), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    close(ev, w2.detach().numpy())
    # backward, scatter flavour and (C = 32 / 64) gather flavour
    refs = dict(xq=t["x_q"].grad, xk=t["x_k"].grad, p1=t["p1"].grad, W3C=t["W3C"].grad, b3C=t["b3C"].grad, gamma=t["gamma"].grad, beta=t["beta"].grad,
                Wa=t["Wa"].grad, ba=t["ba"].grad)
    shapes = dict(xq=(n, C), xk=(n, C), p1=(n, K, 3), W3C=(C, 3), b3C=(C,), gamma=(C,), beta=(C,), Wa=(G, C), ba=(G,))
    flavours = ["scatter"] + (["gather"] if C <= 64 else [])
    for flavour in flavours:
        g = {k: np.full(s, np.nan, np.float32) for k, s in shapes.items()}
        if flavour == "scatter":
            g["xk"][:] = 0                                             # accumulated into: the caller pre-zeroes
            rc = host.cbl_attn_w2_backward(n, K, C, G, P(a["x_q"]), P(a["x_k"]), P(idx), P(a["p1"]), P(a["W3C"]), P(a["b3C"]), P(a["gamma"]), P(a["beta"]),
                                           P(save_mean), P(save_invstd), P(a["Wa"]), P(a["g_w2"]), P(g["xq"]), P(g["xk"]), P(g["p1"]), P(g["W3C"]), P(g["b3C"]),
                                           P(g["gamma"]), P(g["beta"]), P(g["Wa"]), P(g["ba"]), P(ws), ctypes.c_size_t(nbytes), None)
        else:
            inv_start, inv_src = transposed(host, idx)
            rc = host.cbl_attn_w2_backward_csr(n, K, C, G, P(a["x_q"]), P(a["x_k"]), P(idx), P(a["p1"]), P(a["W3C"]), P(a["b3C"]), P(a["gamma"]), P(a["beta"]),
                                               P(save_mean), P(save_invstd), P(a["Wa"]), P(a["g_w2"]), None, P(inv_start), P(inv_src),
                                               P(g["xq"]), P(g["xk"]), P(g["p1"]), P(g["W3C"]), P(g["b3C"]), P(g["gamma"]), P(g["beta"]), P(g["Wa"]), P(g["ba"]),
                                               P(ws), ctypes.c_size_t(nbytes), None)
        assert rc == 0, flavour
        for k in shapes:
            if k == "b3C":                                             # a bias in front of a train-mode BatchNorm has no gradient: what is left is rounding
                assert float(np.abs(g[k]).max()) < 1e-4 * float(refs["W3C"].abs().max())
            else:
                close(g[k], refs[k].numpy())
    if C > 64:
        inv_start, inv_src = transposed(host, idx)
        g = {k: np.zeros(s, np.float32) for k, s in shapes.items()}
        rc = host.cbl_attn_w2_backward_csr(n, K, C, G, P(a["x_q"]), P(a["x_k"]), P(idx), P(a["p1"]), P(a["W3C"]), P(a["b3C"]), P(a["gamma"]), P(a["beta"]),
                                           P(save_mean), P(save_invstd), P(a["Wa"]), P(a["g_w2"]), None, P(inv_start), P(inv_src),
                                           P(g["xq"]), P(g["xk"]), P(g["p1"]), P(g["W3C"]), P(g["b3C"]), P(g["gamma"]), P(g["beta"]), P(g["Wa"]), P(g["ba"]),
                                           P(ws), ctypes.c_size_t(nbytes), None)
        assert rc != 0                                                 # CBL_ERR_UNSUPPORTED for the wide stages: the caller takes the scatter entry


@pytest.mark.parametrize("softmax", [False, True])
@pytest.mark.parametrize("n,K,C", [(300, 16, 32), (260, 8, 64), (150, 16, 128)])
def test_attention_aggregation_pass(host, n, K, C, softmax):
    G = C // 8
    idx, a = scene(n, K, C, seed=3 * C + K)
    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in a.items()}
    ti = torch.from_numpy(idx.astype(np.int64))
    p_r = t["p1"] @ t["W3C"].T + t["b3C"]
    w = torch.softmax(t["logits"], 1) if softmax else t["logits"]
    out = ((t["x_v"][ti] + p_r) * w.repeat(1, 1, 8)).sum(1)            # channel c takes weight c % G (blocks.py:43: view (n, K, 8, G) * w.unsqueeze(2))
    (out * t["g_out"].detach()).sum().backward()
    nbytes = host.cbl_attn_workspace_bytes(C, G)
    ws = np.zeros(nbytes + 64, np.uint8)
    got = np.full((n, C), np.nan, np.float32)
    if softmax:
        weights = np.full((n, K, G), np.nan, np.float32)
        assert host.cbl_attn_agg_softmax_forward(n, K, C, G, P(a["x_v"]), P(idx), P(a["p1"]), P(a["W3C"]), P(a["b3C"]), P(a["logits"]), P(weights), P(got), None) == 0
        close(weights, w.detach().numpy())
    else:
        weights = a["logits"]
        assert host.cbl_attn_agg_forward(n, K, C, G, P(a["x_v"]), P(idx), P(a["p1"]), P(a["W3C"]), P(a["b3C"]), P(weights), P(got), None) == 0
    close(got, out.detach().numpy())
    refs = dict(xv=t["x_v"].grad, p1=t["p1"].grad, W3C=t["W3C"].grad, b3C=t["b3C"].grad, a=t["logits"].grad)
    shapes = dict(xv=(n, C), p1=(n, K, 3), W3C=(C, 3), b3C=(C,), a=(n, K, G))
    for flavour in ["scatter"] + (["gather"] if C <= 64 else []):
        g = {k: np.full(s, np.nan, np.float32) for k, s in shapes.items()}
        if flavour == "scatter":
            g["xv"][:] = 0
            fn = host.cbl_attn_agg_softmax_backward if softmax else host.cbl_attn_agg_backward
            rc = fn(n, K, C, G, P(a["x_v"]), P(idx), P(a["p1"]), P(a["W3C"]), P(a["b3C"]), P(weights), P(a["g_out"]), P(g["xv"]), P(g["p1"]), P(g["W3C"]), P(g["b3C"]),
                    P(g["a"]), P(ws), ctypes.c_size_t(nbytes), None)
        else:
            inv_start, inv_src = transposed(host, idx)
            rc = host.cbl_attn_agg_backward_csr(n, K, C, G, P(a["x_v"]), P(idx), P(a["p1"]), P(a["W3C"]), P(a["b3C"]), P(weights), P(a["g_out"]), None, P(inv_start),
                                                P(inv_src), P(g["xv"]), P(g["p1"]), P(g["W3C"]), P(g["b3C"]), P(g["a"]), P(ws), ctypes.c_size_t(nbytes), 1 if softmax else 0, None)
        assert rc == 0, flavour
        for k in shapes:
            close(g[k], refs[k].numpy())


@pytest.mark.parametrize("rows,C", [(700, 32), (333, 64), (1, 32)])
def test_three_projections_in_one_launch(host, rows, C):
    rng = np.random.default_rng(rows)
    x = rng.normal(size=(rows, C)).astype(np.float32)
    W = [(rng.normal(size=(C, C)) / np.sqrt(C)).astype(np.float32) for _ in range(3)]
    b = [rng.normal(size=C).astype(np.float32) for _ in range(3)]
    gy = [rng.normal(size=(rows, C)).astype(np.float32) for _ in range(3)]

    def arr(xs):
        a = (ctypes.c_void_p * 3)()
        for i, v in enumerate(xs):
            a[i] = None if v is None else v.ctypes.data
        return a
    y = [np.full((rows, C), np.nan, np.float32) for _ in range(3)]
    assert host.cbl_triple_linear_forward(ctypes.c_longlong(rows), C, P(x), arr(W), arr(b), arr(y), None) == 0
    x64 = x.astype(np.float64)
    for p in range(3):
        close(y[p], x64 @ W[p].astype(np.float64).T + b[p], 1e-5)
    gx, gW, gb = np.full((rows, C), np.nan, np.float32), [np.full((C, C), np.nan, np.float32) for _ in range(3)], [np.full(C, np.nan, np.float32) for _ in range(3)]
    nbytes = host.cbl_triple_linear_workspace_bytes(C)
    ws = np.zeros(nbytes + 64, np.uint8)
    assert host.cbl_triple_linear_backward(ctypes.c_longlong(rows), C, P(x), arr(W), arr(gy), P(gx), arr(gW), arr(gb), P(ws), ctypes.c_size_t(nbytes), None) == 0
    close(gx, sum(gy[p].astype(np.float64) @ W[p].astype(np.float64) for p in range(3)), 1e-5)
    for p in range(3):
        close(gW[p], gy[p].astype(np.float64).T @ x64, 1e-5)
        close(gb[p], gy[p].astype(np.float64).sum(0), 1e-5)
