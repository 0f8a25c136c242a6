"""CPU: the product's bucketed furthest point sampling (contrastboundary_amd/csrc/fps_bucket.hip + fps_wave.h, row K2) compiled for the HOST and run with wave
semantics (tests/host_emul/wave: every thread a fibre; v_max_i32_dpp / readlane / ballot / barriers as rendezvous; rocprim's radix sort as std::stable_sort) on
small clouds, against the oracle's restatement of furthestsampling_cuda_kernel (/root/reference/pytorch/lib/pointops/src/sampling/sampling_cuda_kernel.cu:14-129):
the sample SEQUENCES bit for bit, the side effect on `tmp`, the prefix certificate's meaning.  This holds the sample loop's logic — the integer-order maxima, the
holder-lane slot writes, the skipped wave reductions — without a GPU; the `-m gpu` tests hold the device build."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "contrastboundary_amd", "csrc")
EMUL = os.path.join(HERE, "host_emul", "wave")
SRC = os.path.join(HERE, "host_emul", "fps_bucket_host.cpp")
SO = os.path.join(ROOT, "oracle", "_build", "libfps_bucket_host.so")


@pytest.fixture(scope="module")
def host():
    deps = [SRC, os.path.join(CSRC, "fps_bucket.hip"), os.path.join(CSRC, "fps_wave.h"), os.path.join(CSRC, "cbl_common.h"), os.path.join(EMUL, "amdgcn.h"),
            os.path.join(EMUL, "hip", "hip_runtime.h"), os.path.join(EMUL, "rocprim", "device", "device_radix_sort.hpp")]
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, SRC, "-o", SO])
    L = ctypes.CDLL(SO)
    L.host_fps_bucket_workspace_bytes.restype = ctypes.c_size_t
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def ref_bits(n_max):
    """log2 of the reference's block size opt_n_threads(n_max) (cuda_utils.h:11-14): the tie rank's parameter (fps.hip ref_block_threads)"""
    t = min(1 << int(math.log(max(n_max, 1)) / math.log(2.0)), 1024)
    return int(round(math.log2(t)))


def run(L, xyz, off, noff, cert=False):
    xyz, off, noff = O.f32(xyz), O.i32(off), O.i32(noff)
    b, n = len(off), xyz.shape[0]
    n_max = int(np.diff(np.concatenate([[0], off])).max())
    tmp = np.full(n, 1e10, np.float32)
    idx = np.full(int(noff[-1]), -1, np.int32)
    certs = np.full(b, -1, np.int32) if cert else None
    nbytes = L.host_fps_bucket_workspace_bytes(b, n)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = L.host_fps_bucket(b, n, n_max, ref_bits(n_max), P(xyz), P(off), P(noff), P(tmp), P(idx), P(ws), ctypes.c_size_t(nbytes), None, P(certs))
    assert rc == 0
    return idx, tmp, certs, n_max


def cloud(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(0, 1, (n, 3)).astype(np.float32)
    if kind == "surface":                                              # points on two planes and a sphere cap: what a room looks like to the buckets
        u = rng.uniform(0, 1, (n, 2)).astype(np.float32)
        z = np.where(np.arange(n) % 3 == 0, 0.0, np.where(np.arange(n) % 3 == 1, u[:, 0] * 0.3, np.sqrt(np.maximum(0.0, 1 - ((u - 0.5) ** 2).sum(1))))).astype(np.float32)
        return np.concatenate([u, z[:, None]], 1)
    if kind == "lattice":                                              # equal distances everywhere: every arg-max is a tie, the reference's rank decides
        s = int(round(n ** (1 / 3.0))) + 1
        g = np.stack(np.meshgrid(np.arange(s), np.arange(s), np.arange(s), indexing="ij"), -1).reshape(-1, 3)[:n].astype(np.float32) * 0.25
        return g[rng.permutation(n)]
    raise ValueError(kind)


@pytest.mark.parametrize("kind,sizes,ratio", [("uniform", [1500], 4), ("surface", [1800], 4), ("lattice", [1000], 3), ("uniform", [700, 64, 900], 4),
                                              ("surface", [65, 1], 2), ("lattice", [513, 700], 5)])
def test_sample_sequences_equal_the_oracle(host, kind, sizes, ratio):
    xyz = np.concatenate([cloud(kind, n, 10 + i) + 3.0 * i for i, n in enumerate(sizes)])
    off = np.cumsum(sizes)
    noff = np.cumsum([max(1, n // ratio) for n in sizes])
    idx, tmp, _, n_max = run(host, xyz, off, noff)
    ref_idx, ref_tmp = O.furthestsampling(xyz, off, noff, n_max)
    np.testing.assert_array_equal(idx, ref_idx)
    np.testing.assert_array_equal(tmp.view(np.uint32), ref_tmp.view(np.uint32))   # the running distances the reference leaves behind (:56), bit for bit


def test_certificate_counts_leading_unique_maxima(host):
    """cert_out[c] = k means: the first k samples of cloud c were UNIQUE maxima of the running distance (any tie-breaking rule picks them) — checked by replaying the
    sampling in numpy with the kernel's distance expression; only the first half of the samples is tracked"""
    for kind, n, m in (("uniform", 1200, 400), ("lattice", 600, 200)):
        xyz = cloud(kind, n, 5)
        idx, _, certs, _ = run(host, xyz, [n], [m], cert=True)
        ref_idx, _ = O.furthestsampling(xyz, [n], [m])
        np.testing.assert_array_equal(idx, ref_idx)
        k = int(certs[0])
        assert 0 <= k <= (m + 1) // 2
        t = np.full(n, 1e10, np.float32)
        cur = 0
        for j in range(1, max(k, 1)):
            d = xyz - xyz[cur]
            d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)
            t = np.minimum(t, d2)
            best = float(t.max())
            if j < k:
                assert int((t == best).sum()) == 1, (kind, j, k)
            cur = int(np.argmax(t))
            assert cur == int(idx[j])
        if kind == "lattice":
            assert k < 32                                               # a lattice ties within its first samples (the truncated cube's corners are unique maxima)


@pytest.mark.skipif(not os.environ.get("CBL_HOST_EMUL_FULL"), reason="a second (sanitizer) build of the host library: set CBL_HOST_EMUL_FULL=1")
def test_sample_loop_under_address_sanitizer(tmp_path):
    """the same host build with -fsanitize=address in a subprocess: the bucket reads (`sorted`, `rank`: clamped past the cloud's end), the slot array and the
    64-at-a-time index stores of a ragged batch"""
    import sys
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan beside gcc")
    so = os.path.join(ROOT, "oracle", "_build", "libfps_bucket_host_asan.so")
    deps = [SRC, os.path.join(CSRC, "fps_bucket.hip"), os.path.join(CSRC, "fps_wave.h"), os.path.join(CSRC, "cbl_common.h"), os.path.join(EMUL, "amdgcn.h"),
            os.path.join(EMUL, "hip", "hip_runtime.h"), os.path.join(EMUL, "rocprim", "device", "device_radix_sort.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, SRC, "-o", so])
    script = tmp_path / "run.py"
    script.write_text(
        "import ctypes, sys\n"
        "sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import tests.test_fps_bucket_host as T\n"
        "from tests import oracle_lib as O\n"
        "L = ctypes.CDLL(%r)\n"
        "L.host_fps_bucket_workspace_bytes.restype = ctypes.c_size_t\n"
        "sizes = [333, 65, 1]\n"
        "xyz = np.concatenate([T.cloud('uniform', n, i) + 2.0 * i for i, n in enumerate(sizes)])\n"
        "off, noff = np.cumsum(sizes), np.cumsum([111, 30, 1])\n"
        "idx, tmp, certs, n_max = T.run(L, xyz, off, noff, cert=True)\n"
        "ref, _ = O.furthestsampling(xyz, off, noff, n_max)\n"
        "assert np.array_equal(idx, ref)\n"
        "print('ASAN_RUN_DONE')\n" % (ROOT, so))
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "ASAN_RUN_DONE" in r.stdout, (r.returncode, r.stderr[-2000:])
