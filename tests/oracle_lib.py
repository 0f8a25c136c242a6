"""ctypes access to the CPU oracle (oracle/_build/liboracle.so). TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "_build/liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith(".c")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            build_oracle()
        _LIB = ctypes.CDLL(so)
    return _LIB


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---- pointops ---------------------------------------------------------------------------------
def knnquery(nsample, xyz, new_xyz, offset, new_offset, q0=0, q1=None):
    xyz, new_xyz, offset, new_offset = f32(xyz), f32(new_xyz), i32(offset), i32(new_offset)
    m = new_xyz.shape[0]
    q1 = m if q1 is None else q1
    idx = np.zeros((m, nsample), np.int32)
    d2 = np.zeros((m, nsample), np.float32)
    lib().oracle_knnquery_range(q0, q1, nsample, P(xyz), P(new_xyz), P(offset), P(new_offset), P(idx), P(d2))
    return idx, d2


def furthestsampling(xyz, offset, new_offset, n_max=None):
    xyz, offset, new_offset = f32(xyz), i32(offset), i32(new_offset)
    if n_max is None:
        n_max = int(np.max(np.diff(np.concatenate([[0], offset]))))
    tmp = np.full((xyz.shape[0],), 1e10, np.float32)
    idx = np.zeros((int(new_offset[-1]),), np.int32)
    lib().oracle_furthestsampling(len(offset), int(n_max), P(xyz), P(offset), P(new_offset), P(tmp), P(idx))
    return idx, tmp


def grouping_forward(inp, idx):
    inp, idx = f32(inp), i32(idx)
    m, ns = idx.shape
    c = inp.shape[1]
    out = np.empty((m, ns, c), np.float32)
    lib().oracle_grouping_forward(m, ns, c, P(inp), P(idx), P(out))
    return out


def grouping_backward(grad_out, idx, n):
    grad_out, idx = f32(grad_out), i32(idx)
    m, ns, c = grad_out.shape
    gi = np.zeros((n, c), np.float32)
    lib().oracle_grouping_backward(m, ns, c, P(grad_out), P(idx), P(gi))
    return gi


def interpolation_forward(inp, idx, w):
    inp, idx, w = f32(inp), i32(idx), f32(w)
    n, k = idx.shape
    c = inp.shape[1]
    out = np.zeros((n, c), np.float32)
    lib().oracle_interpolation_forward(n, c, k, P(inp), P(idx), P(w), P(out))
    return out


def interpolation_backward(go, idx, w, m):
    go, idx, w = f32(go), i32(idx), f32(w)
    n, k = idx.shape
    c = go.shape[1]
    gi = np.zeros((m, c), np.float32)
    lib().oracle_interpolation_backward(n, c, k, P(go), P(idx), P(w), P(gi))
    return gi


def subtraction_forward(a, b, idx):
    a, b, idx = f32(a), f32(b), i32(idx)
    n, ns = idx.shape
    c = a.shape[1]
    out = np.zeros((n, ns, c), np.float32)
    lib().oracle_subtraction_forward(n, ns, c, P(a), P(b), P(idx), P(out))
    return out


def subtraction_backward(idx, go):
    idx, go = i32(idx), f32(go)
    n, ns, c = go.shape
    g1 = np.zeros((n, c), np.float32)
    g2 = np.zeros((n, c), np.float32)
    lib().oracle_subtraction_backward(n, ns, c, P(idx), P(go), P(g1), P(g2))
    return g1, g2


def aggregation_forward(inp, pos, w, idx):
    inp, pos, w, idx = f32(inp), f32(pos), f32(w), i32(idx)
    n, ns, c = pos.shape
    wc = w.shape[2]
    out = np.zeros((n, c), np.float32)
    lib().oracle_aggregation_forward(n, ns, c, wc, P(inp), P(pos), P(w), P(idx), P(out))
    return out


def aggregation_backward(inp, pos, w, idx, go):
    inp, pos, w, idx, go = f32(inp), f32(pos), f32(w), i32(idx), f32(go)
    n, ns, c = pos.shape
    wc = w.shape[2]
    gi = np.zeros((inp.shape[0], c), np.float32)
    gp = np.zeros((n, ns, c), np.float32)
    gw = np.zeros((n, ns, wc), np.float32)
    lib().oracle_aggregation_backward(n, ns, c, wc, P(inp), P(pos), P(w), P(idx), P(go), P(gi), P(gp), P(gw))
    return gi, gp, gw


# ---- TF-side ops -------------------------------------------------------------------------------
def grid_subsampling(points, lengths, dl):
    points, lengths = f32(points), i32(lengths)
    n, b = points.shape[0], len(lengths)
    out = np.zeros((n, 3), np.float32)
    ol = np.zeros(b, np.int32)
    m = lib().oracle_batch_grid_subsampling(n, P(points), b, P(lengths), ctypes.c_float(dl), P(out), P(ol))
    return out[:m].copy(), ol


def grid_subsampling_full(points, features, labels, dl):
    points, features, labels = f32(points), f32(features), i32(labels)
    n, fdim, ldim = points.shape[0], features.shape[1], labels.shape[1]
    op = np.zeros((n, 3), np.float32); of = np.zeros((n, fdim), np.float32); ol = np.zeros((n, ldim), np.int32); tie = np.zeros((n, ldim), np.int32)
    m = lib().oracle_grid_subsampling_full(n, P(points), fdim, P(features), ldim, P(labels), ctypes.c_float(dl), P(op), P(of), P(ol), P(tie))
    return op[:m].copy(), of[:m].copy(), ol[:m].copy(), tie[:m].copy()


def radius_neighbors(queries, supports, q_len, s_len, radius, limit):
    queries, supports, q_len, s_len = f32(queries), f32(supports), i32(q_len), i32(s_len)
    nq, ns = queries.shape[0], supports.shape[0]
    out = np.zeros((nq, limit), np.int32); counts = np.zeros(nq, np.int32)
    mc = lib().oracle_radius_neighbors(nq, P(queries), ns, P(supports), len(q_len), P(q_len), P(s_len), ctypes.c_float(radius), limit, P(out), P(counts))
    return out, counts, mc


def knn_batch(points, queries, k):
    points, queries = f32(points), f32(queries)
    B, N, _ = points.shape
    M = queries.shape[1]
    out = np.zeros((B, M, k), np.int64)
    lib().oracle_knn_batch(B, N, M, k, P(points), P(queries), P(out))
    return out


# ---- oracle/_ref: the reference's own TF-side C++ cores (built in the build container, prebuilt .so travels) ------
_REF = {}


def ref(name):
    """ctypes handle of oracle/_ref/lib<name>.so or None if it is not available"""
    if name not in _REF:
        so = os.path.join(ORACLE_DIR, "_ref", f"lib{name}.so")
        if not os.path.exists(so) and os.path.isdir("/root/reference"):
            subprocess.call(["make", "-s", "-C", ORACLE_DIR, "ref"])
        _REF[name] = ctypes.CDLL(so) if os.path.exists(so) else None
    return _REF[name]
