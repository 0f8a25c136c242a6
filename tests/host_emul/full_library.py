"""TEST INFRASTRUCTURE: the WHOLE product library (every contrastboundary_amd/csrc/*.hip except version.hip, whose two functions query the device) compiled for the
HOST with wave semantics (tests/host_emul/wave) into oracle/_build/libcbl_amd_host.so — one translation unit per kernel file, as in the product build, so the
composite entry points that call across files (cbl_pyramid, cbl_convnet_step) run on a CPU exactly as the device library issues them.  Not product code: nothing
outside tests/ loads it."""
import ctypes
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "contrastboundary_amd", "csrc")
EMUL = os.path.join(HERE, "wave")
GEN = os.path.join(HERE, "host_tu.py")
BUILD = os.path.join(ROOT, "oracle", "_build", "full")
SO = os.path.join(BUILD, "libcbl_amd_host.so")
SO_ASAN = os.path.join(BUILD, "asan", "libcbl_amd_host.so")       # the same build with -fsanitize=address (CBL_FULL_LIBRARY_ASAN=1, under LD_PRELOAD=libasan)
SKIPPED = {"version"}                                                 # cbl_version / cbl_device_arch_ok: a device query, nothing to emulate
WHOLE = {"pt_layer"}                                                  # files whose `#ifndef CBL_HOST_WAVE_EMULATION` sections this build includes


def sources():
    return sorted(f for f in glob.glob(os.path.join(CSRC, "*.hip")) if os.path.splitext(os.path.basename(f))[0] not in SKIPPED)


def build(asan=False):
    """-> path of the host library, rebuilt when any kernel file, header or piece of the emulator is newer"""
    so, out = (SO_ASAN, os.path.join(BUILD, "asan")) if asan else (SO, BUILD)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [GEN, os.path.abspath(__file__), os.path.join(ROOT, "include", "cbl_amd.h")]
    deps += [os.path.join(d, f) for d, _, fs in os.walk(EMUL) for f in fs]
    if os.path.exists(so) and os.path.getmtime(so) >= max(os.path.getmtime(d) for d in deps):
        return so
    os.makedirs(out, exist_ok=True)
    extra = ["-g", "-fsanitize=address"] if asan else []

    def one(src):
        name = os.path.splitext(os.path.basename(src))[0]
        tu, obj = os.path.join(out, name + ".cpp"), os.path.join(out, name + ".o")
        subprocess.check_call([sys.executable, GEN] + (["--whole"] if name in WHOLE else []) + [tu, src])
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-c", "-ffp-contract=off", "-Wno-unknown-pragmas"] + extra +
                              ["-I" + EMUL, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, tu, "-o", obj])
        return obj
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(one, sources()))
    tmp = so + ".%d.tmp" % os.getpid()
    subprocess.check_call(["g++", "-shared"] + extra + objs + ["-o", tmp])
    os.replace(tmp, so)
    return so


def load():
    L = ctypes.CDLL(build(asan=bool(os.environ.get("CBL_FULL_LIBRARY_ASAN"))))
    for name in ("cbl_radius_neighbors_workspace_bytes", "cbl_pyramid_layer_workspace_bytes", "cbl_convnet_step_workspace_bytes"):
        getattr(L, name).restype = ctypes.c_size_t
    return L
