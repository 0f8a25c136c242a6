"""CPU: the WHOLE product library compiled for the host with wave semantics (tests/host_emul/full_library.py: one translation unit per kernel file, as the
product build) and its composite entry points run as the device library issues them:
  - every function include/cbl_amd.h declares is defined by the host build (all kernel files compile for the host; only the device query of version.hip is left out);
  - the ConvNet step of one scene: `cbl_pyramid` (/root/reference/tensorflow/datasets/base.py:767-842) followed by `cbl_convnet_step` (AdaptiveWeight forward +
    backward, local_aggregation_operators.py:360-484; scene labels through the pools, heads/head.py:25-49; contrast head forward + backward, head.py:462-807) —
    what bench.py's ConvNet leg times per scene — against the oracles layer by layer, with the tolerances tests/test_gpu_bench_convnet.py uses on the device;
  - the test loop's accumulation of per-crop predictions (`cbl_cumulate_probs`, /root/reference/pytorch/tool/test.py:330-352) against numpy's indexed assignment."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from contrastboundary_amd import convnet_path as CP
from oracle import cbl_oracle as C, local_aggregation_oracle as LA
from tests import oracle_lib as O
from tests.host_emul import full_library


@pytest.fixture(scope="module")
def host():
    return full_library.load()


def P(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def test_host_build_defines_every_declared_entry_point(host):
    header = open(full_library.ROOT + "/include/cbl_amd.h").read()
    declared = set(re.findall(r"\b(cbl_[a-z0-9_]+)\s*\(", header))
    out = subprocess.run(["nm", "-D", "--defined-only", full_library.SO], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in out.splitlines()}
    assert len(declared) > 100
    assert declared - defined == {"cbl_version", "cbl_device_arch_ok"}


class Layer(ctypes.Structure):
    """CblConvnetLayer (include/cbl_amd.h)"""
    _fields_ = [(k, ctypes.c_int) for k in ("n", "K", "C", "Kp", "d")] + [("radius", ctypes.c_float)] + \
               [(k, ctypes.c_void_p) for k in ("points", "neighbors", "features", "fc_weight", "fc_bias", "grad_out", "latent", "pools", "aw_out", "grad_features",
                                               "grad_fc_weight", "grad_fc_bias", "cbl_loss", "cbl_mask", "grad_latent", "labels")]


def ptrs(arrs, count):
    a = (ctypes.c_void_p * count)()
    for i, x in enumerate(arrs):
        a[i] = x.ctypes.data
    return a


def test_convnet_step_of_one_scene_against_the_oracles(host):
    layers, widths, n = 3, [72, 16, 144], 4200
    a = CP.ConvNetScene.synthetic_numpy(n, seed=4, b=2, layers=layers, widths=widths)
    xyz, lens = a["points"], a["lengths"]
    b = len(lens)
    r0, dl0 = CP.DL0 * CP.DENSITY / 2.0, CP.DL0
    limits = np.int32(CP.LIMITS[:layers])
    # ---- the pyramid in one call
    grid_bytes = host.cbl_radius_neighbors_workspace_bytes(b, n)
    grids = [np.zeros(grid_bytes + 64, np.uint8) for _ in range(layers)]
    nb = [np.full((n, int(limits[l])), -5, np.int32) for l in range(layers)]
    pp = [np.full((n, 3), np.nan, np.float32) for _ in range(layers - 1)]
    pl = [np.full(b, -1, np.int32) for _ in range(layers - 1)]
    po = [np.full((n, int(limits[l])), -5, np.int32) for l in range(layers - 1)]
    up = [np.full((n, int(limits[l])), -5, np.int32) for l in range(layers - 1)]
    mx, sizes = np.full(3 * layers, -1, np.int32), np.full(layers, -1, np.int32)
    nbytes = host.cbl_pyramid_layer_workspace_bytes(b, n)
    ws = np.zeros(nbytes + 64, np.uint8)
    rc = host.cbl_pyramid(b, n, P(xyz), P(lens), ctypes.c_float(r0), ctypes.c_float(dl0), layers, P(limits), ptrs(grids, layers), ctypes.c_size_t(grid_bytes),
                          ptrs(nb, layers), ptrs(pp, layers), ptrs(pl, layers), ptrs(po, layers), ptrs(up, layers), P(mx), P(sizes), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    # ---- the layers in one call, on the tables cropped to the widths the reference's dynamic shapes give them (datasets/base.py:756-765)
    pts = [xyz] + [np.ascontiguousarray(pp[l][:sizes[l + 1]]) for l in range(layers - 1)]
    tables = [np.ascontiguousarray(nb[l][:sizes[l], :min(int(mx[3 * l]), int(limits[l]))]) for l in range(layers)]
    pools = [None] + [np.ascontiguousarray(po[l][:sizes[l + 1], :min(int(mx[3 * l + 1]), int(limits[l]))]) for l in range(layers - 1)]
    arrs = [CP.ConvNetScene.layer_arrays_numpy(a["seeds"][l], int(sizes[l]), widths[l]) for l in range(layers)]
    outs, structs = [], (Layer * layers)()
    for l in range(layers):
        m, c = int(sizes[l]), widths[l]
        o = dict(aw_out=np.full((m, c), np.nan, np.float32), grad_features=np.full((m, c), np.nan, np.float32), grad_fc_weight=np.full((3, c), np.nan, np.float32),
                 grad_fc_bias=np.full(c, np.nan, np.float32), cbl_loss=np.full(1, np.nan, np.float32), cbl_mask=np.full(m, -1, np.int32),
                 grad_latent=np.full((m, CP.CBL_DIM), np.nan, np.float32), labels=np.full(m, -1, np.int32))
        outs.append(o)
        s = structs[l]
        s.n, s.K, s.C, s.Kp, s.d, s.radius = m, tables[l].shape[1], c, (pools[l].shape[1] if l else 0), CP.CBL_DIM, r0 * 2 ** l
        s.points, s.neighbors, s.features = pts[l].ctypes.data, tables[l].ctypes.data, arrs[l]["feat"].ctypes.data
        s.fc_weight, s.fc_bias, s.grad_out, s.latent = a["fc_weight"][l].ctypes.data, a["fc_bias"][l].ctypes.data, arrs[l]["grad"].ctypes.data, arrs[l]["latent"].ctypes.data
        s.pools = pools[l].ctypes.data if l else None
        for k, v in o.items():
            setattr(s, k, v.ctypes.data)
    labels64 = np.ascontiguousarray(a["labels"], np.int64)
    nbytes = host.cbl_convnet_step_workspace_bytes(layers, structs, CP.NUM_CLASSES)
    assert nbytes > 0
    ws = np.zeros(nbytes + 64, np.uint8)
    assert host.cbl_convnet_step(layers, structs, P(labels64), CP.NUM_CLASSES, ctypes.c_float(1.0), ctypes.c_float(0.1), P(ws), ctypes.c_size_t(nbytes - 256), None) != 0
    rc = host.cbl_convnet_step(layers, structs, P(labels64), CP.NUM_CLASSES, ctypes.c_float(1.0), ctypes.c_float(0.1), P(ws), ctypes.c_size_t(nbytes), None)
    assert rc == 0
    # ---- the oracles, layer by layer
    p, ln, r, dl, lab = xyz, lens, r0, dl0, a["labels"]
    for l in range(layers):
        assert int(sizes[l]) == p.shape[0]
        lim = int(limits[l])
        ref, _, mc = O.radius_neighbors(p, p, ln, ln, r, lim)
        t = ref[:, :min(mc, lim)]
        np.testing.assert_array_equal(tables[l], t)
        W, bias, o = a["fc_weight"][l], a["fc_bias"][l], outs[l]
        out = LA.adaptive_weight(p, p, t, arrs[l]["feat"], r, W, bias, "mean")
        np.testing.assert_allclose(o["aw_out"], out, rtol=1e-4, atol=1e-4 * np.abs(out).max())
        gf, gw, gb = LA.adaptive_weight_grads(p, p, t, arrs[l]["feat"], r, W, bias, arrs[l]["grad"], "mean")
        np.testing.assert_allclose(o["grad_features"], gf, rtol=1e-4, atol=1e-4 * np.abs(gf).max())
        np.testing.assert_allclose(o["grad_fc_weight"], gw, rtol=1e-4, atol=1e-4 * np.abs(gw).max())
        np.testing.assert_allclose(o["grad_fc_bias"], gb, rtol=1e-4, atol=1e-4 * np.abs(gb).max())
        np.testing.assert_array_equal(o["labels"], lab)
        rl, rg, rm = C.tf_contrast(arrs[l]["latent"], lab, t, temperature=1.0, weight=0.1)
        assert abs(float(o["cbl_loss"][0]) - rl) < 1e-4 * max(1.0, abs(rl))
        np.testing.assert_array_equal(o["cbl_mask"] > 0, rm > 0)
        assert (rm > 0).any() and not (rm > 0).all()
        np.testing.assert_allclose(o["grad_latent"], rg, rtol=1e-4, atol=1e-4 * max(np.abs(rg).max(), 1e-12))
        if l == layers - 1:
            break
        sub, sl = O.grid_subsampling(p, ln, 2 * dl)
        np.testing.assert_array_equal(pts[l + 1].view(np.uint32), sub.view(np.uint32))
        refp, _, mcp = O.radius_neighbors(sub, p, sl, ln, r, lim)
        pt = refp[:, :min(mcp, lim)]
        np.testing.assert_array_equal(pools[l + 1], pt)
        lab = C.tf_scene_label(lab, pt, CP.NUM_CLASSES, "max")
        p, ln, r, dl = np.ascontiguousarray(sub), sl.astype(np.int32), 2 * r, 2 * dl


@pytest.mark.parametrize("mode,smooth", [(0, 0.0), (1, 0.95), (2, 0.0)])
def test_cumulate_probs_keeps_the_last_row_of_a_duplicated_point(host, mode, smooth):
    """tool/test.py:330-352 on the CPU: `probs[inds] += pred` is gather, add, indexed ASSIGNMENT — of rows that share a point the last one counts"""
    rng = np.random.default_rng(7)
    n, ncls, m = 900, 13, 2500                                        # crops overlap: most points appear more than once; some never; some indices outside
    inds = rng.integers(0, n - 50, m).astype(np.int64)
    inds[::97] = n + 3; inds[5] = -1                                  # rows the kernels skip (the Python mirror never passes them: bounds are the caller's)
    pred = rng.normal(size=(m, ncls)).astype(np.float32)
    probs = rng.normal(size=(n, ncls)).astype(np.float32)
    ref = probs.copy()
    ok = (inds >= 0) & (inds < n)
    i, q = inds[ok], pred[ok]
    if mode == 0:
        ref[i] = ref[i] + q
    elif mode == 1:
        ref[i] = np.float32(smooth) * ref[i] + (np.float32(1.0) - np.float32(smooth)) * q
    else:
        ref[i] = q
    scratch = np.zeros(n, np.int32)
    rc = host.cbl_cumulate_probs(n, ncls, m, P(inds), P(pred), ctypes.c_float(smooth), mode, P(probs), P(scratch), None)
    assert rc == 0
    np.testing.assert_array_equal(probs.view(np.uint32), ref.view(np.uint32))


@pytest.mark.skipif(not os.environ.get("CBL_HOST_EMUL_FULL"), reason="a sanitizer build of the whole library and a quarter of an hour of emulation: set CBL_HOST_EMUL_FULL=1")
def test_whole_library_cases_under_address_sanitizer():
    """every test module over the whole-library host build again, against the same build with -fsanitize=address (operands are numpy buffers of exactly their logical
    sizes, `__shared__` arrays static arrays): the clamped loads, masked stores and workspace carving of every entry point.  Last run clean at the commit that
    added this test (14 minutes)."""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan beside gcc")
    env = dict(os.environ, CBL_FULL_LIBRARY_ASAN="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    env.pop("CBL_HOST_EMUL_FULL")                                    # (not this test again)
    mods = ["tests/test_heads_host.py", "tests/test_gather_variants_host.py", "tests/test_full_library_host.py", "tests/test_attention_host.py", "tests/test_knn_variants_host.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + mods, capture_output=True, text=True, timeout=3600, env=env, cwd=full_library.ROOT)
    assert "AddressSanitizer" not in r.stdout + r.stderr, (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
