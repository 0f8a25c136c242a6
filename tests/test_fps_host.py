"""CPU: the product's furthest point sampling as its ENTRY POINTS see it (contrastboundary_amd/csrc/fps.hip: the register / LDS-resident kernel for small clouds, the
dispatch to the bucketed kernel of fps_bucket.hip above 3072 points, and the chain of samplings with prefix certificates — the network samples
40960 -> 10240 -> 2560 -> 640 -> 160, /root/reference/pytorch/model/blocks.py:61-68) compiled for the HOST and run with wave semantics (tests/host_emul/wave),
against the oracle's restatement of furthestsampling_cuda_kernel (sampling_cuda_kernel.cu:14-129): sample sequences bit for bit, stage by stage of a chain."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import oracle_lib as O
from tests.test_fps_bucket_host import cloud

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "contrastboundary_amd", "csrc")
EMUL = os.path.join(HERE, "host_emul", "wave")
GEN = os.path.join(HERE, "host_emul", "host_tu.py")
BUILD = os.path.join(ROOT, "oracle", "_build")
SO = os.path.join(BUILD, "libfps_host.so")
FILES = ["fps", "fps_bucket", "knn_grid", "knn_exact", "knn_select", "knn_dispatch"]       # fps_bucket's bounding boxes come from knn_grid.hip


@pytest.fixture(scope="module")
def host():
    srcs = [os.path.join(CSRC, f + ".hip") for f in FILES]
    deps = srcs + [GEN, os.path.abspath(__file__), os.path.join(CSRC, "fps_wave.h"), os.path.join(CSRC, "cbl_common.h"), os.path.join(EMUL, "amdgcn.h"),
                   os.path.join(EMUL, "hip", "hip_runtime.h"), os.path.join(EMUL, "rocprim", "device", "device_radix_sort.hpp")]
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        objs = []
        for f, src in zip(FILES, srcs):
            tu, obj = os.path.join(BUILD, f + "_host.cpp"), os.path.join(BUILD, f + "_fpshost.o")
            subprocess.check_call([sys.executable, GEN, tu, src])
            subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-c", "-ffp-contract=off", "-Wno-unknown-pragmas",
                                   "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, tu, "-o", obj])
            objs.append(obj)
        subprocess.check_call(["g++", "-shared"] + objs + ["-o", SO])
    L = ctypes.CDLL(SO)
    L.cbl_furthestsampling_workspace_bytes.restype = ctypes.c_size_t
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def sample(L, xyz, off, noff, cert_in=None, want_cert=False):
    xyz, off, noff = O.f32(xyz), O.i32(off), O.i32(noff)
    b, n = len(off), xyz.shape[0]
    n_max = int(np.diff(np.concatenate([[0], off])).max())
    tmp, idx = np.full(n, 1e10, np.float32), np.full(int(noff[-1]), -1, np.int32)
    nbytes = L.cbl_furthestsampling_workspace_bytes(b, n, n_max)
    ws = np.zeros(nbytes + 64, np.uint8)
    cert_out = np.full(b, -1, np.int32) if want_cert else None
    if want_cert or cert_in is not None:
        rc = L.cbl_furthestsampling_chain(b, n, n_max, P(xyz), P(off), P(noff), P(tmp), P(idx), P(cert_in), P(cert_out), P(ws), ctypes.c_size_t(nbytes), None)
    else:
        rc = L.cbl_furthestsampling_ws(b, n, n_max, P(xyz), P(off), P(noff), P(tmp), P(idx), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    return idx, cert_out, n_max


@pytest.mark.parametrize("kind,sizes,ratio", [("uniform", [900], 4), ("lattice", [700], 3), ("uniform", [300, 1, 1500], 4), ("surface", [2500], 5), ("uniform", [3100], 8)])
def test_sampling_entry_equals_the_oracle(host, kind, sizes, ratio):
    """clouds below 3072 points take the resident kernel of fps.hip (1, 4 rows of 1024 lanes ...), the last case the bucketed one through the dispatcher"""
    xyz = np.concatenate([cloud(kind, n, 30 + i) + 3.0 * i for i, n in enumerate(sizes)])
    off, noff = np.cumsum(sizes), np.cumsum([max(1, n // ratio) for n in sizes])
    idx, _, n_max = sample(host, xyz, off, noff)
    ref, _ = O.furthestsampling(xyz, off, noff, n_max)
    np.testing.assert_array_equal(idx, ref)


def test_a_chain_of_samplings_with_prefix_certificates(host):
    """3200 -> 800 -> 200 -> 50 as the TransitionDown stages run it: every stage samples the previous stage's samples IN SAMPLING ORDER; with a certificate the later
    stages are answered as prefixes.  Every stage equals the oracle run on that stage's input."""
    xyz = cloud("uniform", 3200, 77)
    off = np.int32([3200])
    cert = None
    for m in (800, 200, 50):
        noff = np.int32([m])
        idx, cert, n_max = sample(host, xyz, off, noff, cert_in=cert, want_cert=True)
        ref, _ = O.furthestsampling(xyz, off, noff, n_max)
        np.testing.assert_array_equal(idx, ref)
        assert 0 <= int(cert[0])
        xyz, off = np.ascontiguousarray(xyz[idx]), noff
