"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/cbl_amd.h declares.
No compute call is made (no GPU here)."""
import ctypes
import os

import pytest

from contrastboundary_amd import _lib, build


def test_library_builds_and_exports_every_declared_symbol():
    so = build.build()
    assert os.path.exists(so)
    L = ctypes.CDLL(so)
    syms = _lib.declared_symbols()
    assert len(syms) >= 16
    missing = [s for s in syms if not hasattr(L, s)]
    assert missing == []
    assert b"gfx950" in _lib.lib().cbl_version()


def test_code_object_is_gfx950_only():
    # the fat binary inside the .so must carry exactly one device target: gfx950 (no multi-arch / fallback paths)
    import re
    data = open(build.SO, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", data))     # offload-bundle entry ids
    assert targets == {b"gfx950"}, targets


def test_bad_arguments_are_rejected_without_a_gpu():
    L = _lib.lib()
    null = ctypes.c_void_p(0)
    # argument validation happens before any HIP call, so it can be exercised on the CPU box
    assert L.cbl_knnquery_exact(1, 10, 10, 0, null, null, null, null, null, null, null) == -1          # nsample = 0
    assert L.cbl_knnquery_exact(1, 10, 10, 2000, null, null, null, null, null, null, null) == -1       # nsample > 1024
    assert L.cbl_knnquery_exact(1, 10, 10, 4, null, null, null, null, null, null, null) == -1          # null pointers
    assert L.cbl_grouping_forward(-1, 4, 4, null, null, null, null) == -1
    assert L.cbl_grouping_forward(0, 4, 4, null, null, null, null) == 0                                 # empty = no-op
    assert L.cbl_aggregation_forward(4, 4, 4, 0, null, null, null, null, null, null) == -1             # w_c = 0


def test_host_mirror_refuses_cpu_tensors():
    import torch
    from contrastboundary_amd import pointops
    x = torch.zeros(8, 3)
    o = torch.tensor([8], dtype=torch.int32)
    with pytest.raises(_lib.CblError):
        pointops.knnquery(2, x, x, o, o)
    with pytest.raises(_lib.CblError):
        pointops.grouping(torch.zeros(8, 4), torch.zeros(8, 2, dtype=torch.int32))


def test_dropin_has_no_cpu_path():
    """the drop-in module loads without a GPU (ctypes) and refuses CPU tensors loudly instead of handing their addresses to a kernel"""
    import sys
    import pytest
    import torch
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "contrastboundary_amd", "dropin")
    sys.path.insert(0, d)
    try:
        import pointops_cuda
        xyz = torch.rand(64, 3); off = torch.tensor([64], dtype=torch.int32)
        with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
            pointops_cuda.knnquery_cuda(64, 4, xyz, xyz, off, off, torch.zeros(64, 4, dtype=torch.int32), torch.zeros(64, 4))
    finally:
        sys.path.remove(d)


def test_every_module_of_the_package_imports():
    """a syntax error in a module only the GPU tests import must not wait for the GPU box to be seen"""
    import importlib
    import pkgutil
    import contrastboundary_amd
    names = [m.name for m in pkgutil.walk_packages(contrastboundary_amd.__path__, "contrastboundary_amd.")]
    assert len(names) >= 15, names
    for name in names:
        importlib.import_module(name)
    for name in ("bench", "__graft_entry__"):
        importlib.import_module(name)
