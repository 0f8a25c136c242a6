"""CPU: the product's pipelined gather kernel (contrastboundary_amd/csrc/query_group_pipe.h: persistent waves, ids a trip ahead, next rows requested
before the stores, trips past the end re-storing piece 0) compiled for the HOST and run with wave semantics (tests/host_emul/wave) against numpy —
QueryAndGroup, /root/reference/pytorch/lib/pointops/functions/pointops.py:79-100: out[i, k] = (xyz[idx[i, k]] - xyz[i], feat[idx[i, k]]), bit for bit,
for every grid size from one workgroup (many trips per wave) to more workgroups than pieces (waves with nothing but trips past the end)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "contrastboundary_amd", "csrc")
EMUL = os.path.join(HERE, "host_emul", "wave")
SRC = os.path.join(HERE, "host_emul", "gather_pipe_host.cpp")
SO = os.path.join(ROOT, "oracle", "_build", "libgather_pipe_host.so")


@pytest.fixture(scope="module")
def host():
    deps = [SRC, os.path.join(CSRC, "query_group_pipe.h"), os.path.join(CSRC, "cbl_common.h"), os.path.join(EMUL, "gather_wave.h"), os.path.join(EMUL, "hip", "hip_runtime.h")]
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-Wno-attributes",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, SRC, "-o", SO])
    return ctypes.CDLL(SO)


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("c,k", [(32, 8), (32, 16), (64, 8), (64, 16)])
@pytest.mark.parametrize("n,grid", [(37, 1), (37, 8), (37, 16), (130, 2), (9, 24)])
def test_pipelined_gather_on_the_host_is_the_reference_grouping(host, c, k, n, grid):
    rng = np.random.default_rng(1000 * c + 10 * k + n)
    xyz = rng.normal(size=(n, 3)).astype(np.float32)
    feat = rng.normal(size=(n, c)).astype(np.float32)
    idx = rng.integers(0, n, size=(n, k)).astype(np.int32)
    order = rng.permutation(n).astype(np.int32)
    out = np.full((n, k, 3 + c), np.nan, np.float32)
    fn = getattr(host, "gather_pipe_c%d_k%d" % (c, k))
    assert fn(ctypes.c_uint(grid), ctypes.c_uint(n), P(xyz), P(xyz), P(feat), P(idx), P(order), P(out)) == 0
    want = np.concatenate([xyz[idx] - xyz[:, None, :], feat[idx]], axis=-1)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
