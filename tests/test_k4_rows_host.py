"""CPU: the product's K4 kernel (contrastboundary_amd/csrc/k4_rows_pipe.h — grouping backward as a gather over the transposed neighbour table, list bounds
and pair ids prefetched a target ahead, two batches of eight rows in flight) compiled for the HOST and run with wave semantics (tests/host_emul/wave)
against the reference loop (/root/reference/pytorch/lib/pointops/src/grouping/grouping_cuda_kernel.cu:16-25) run sequentially in float32: the sum over a
target's pairs in ascending pair order, bit for bit — lists of 0 ... 150 pairs (past the 64 a lane group prefetches), sliced rows (stride 3 + c, offset 3)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "contrastboundary_amd", "csrc")
EMUL = os.path.join(HERE, "host_emul", "wave")
SRC = os.path.join(HERE, "host_emul", "k4_rows_host.cpp")
SO = os.path.join(ROOT, "oracle", "_build", "libk4_rows_host.so")


@pytest.fixture(scope="module")
def host():
    deps = [SRC, os.path.join(CSRC, "k4_rows_pipe.h"), os.path.join(CSRC, "cbl_common.h"), os.path.join(EMUL, "gather_wave.h"), os.path.join(EMUL, "hip", "hip_runtime.h")]
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-Wno-attributes",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, SRC, "-o", SO])
    return ctypes.CDLL(SO)


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("n,K,c,grid,ordered", [(50, 16, 64, 1, True), (50, 16, 64, 8, False), (23, 8, 128, 16, True), (200, 4, 64, 3, True), (12, 36, 64, 40, False)])
def test_k4_gather_on_the_host_is_the_sequential_reference_sum(host, n, K, c, grid, ordered):
    rng = np.random.default_rng(n * 100 + K)
    m = n                                                                       # queries = supports here
    idx = rng.integers(0, n, size=(m, K)).astype(np.int32)
    idx[: m // 3] = rng.integers(0, 2, size=(m // 3, K))                         # two targets with very long lists (> 64 pairs when m K is large enough)
    stride, off = c + 3, 3
    go = rng.normal(size=(m * K, stride)).astype(np.float32)
    flat = idx.reshape(-1)
    src = np.argsort(flat, kind="stable").astype(np.int32)                      # pairs grouped by target, ascending pair id inside a target
    counts = np.bincount(flat, minlength=n)
    first = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    ranks = rng.permutation(n).astype(np.int32) if ordered else np.arange(n, dtype=np.int32)
    inv_start = np.zeros(n + 1, np.int32); inv_src = np.zeros(max(m * K, 1), np.int32)
    o = 0
    for r, j in enumerate(ranks):
        inv_start[r] = o
        seg = src[first[j]:first[j + 1]]
        inv_src[o:o + len(seg)] = seg
        o += len(seg)
    inv_start[n] = o
    gi = np.full((n, c), np.nan, np.float32)
    assert host.k4_rows(ctypes.c_uint(grid), ctypes.c_uint(n), c, stride, off, P(go), P(ranks) if ordered else None, P(inv_start), P(inv_src), P(gi)) == 0
    want = np.zeros((n, c), np.float32)
    for p in range(m * K):                                                      # the reference loop, one pair after the other (float32 adds in pair order)
        want[flat[p]] = want[flat[p]] + go[p, off:off + c]
    assert counts.max() > 64 or m * K < 200
    assert np.array_equal(gi.view(np.uint32), want.view(np.uint32))
