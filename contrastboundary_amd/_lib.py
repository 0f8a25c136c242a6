"""ctypes binding of libcbl_amd.so (include/cbl_amd.h).  There is NO fallback: if the HIP library is
missing and cannot be built, or a call returns an error code, this raises."""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_PKG, "lib", "libcbl_amd.so")
_HEADER = os.path.join(os.path.dirname(_PKG), "include", "cbl_amd.h")
_lib = None


class CblError(RuntimeError):
    pass


ERR_UNSUPPORTED = -3            # CBL_ERR_UNSUPPORTED (cbl_amd.h)


def declared_symbols():
    """every function name include/cbl_amd.h declares"""
    txt = open(_HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cbl_\w+)\s*\(", txt)))


def lib():
    global _lib
    if _lib is None:
        # torch first: its wheel bundles its own libamdhip64 (soname libamdhip64.so.7).  Loaded first, it is
        # the ONE HIP runtime of the process and libcbl_amd.so binds to it by soname; loaded second, the
        # process would hold two runtimes and torch's streams/pointers would be foreign to ours.
        import torch  # noqa: F401
        from . import build as _build
        if _build.is_stale():
            _build.build()          # raises if hipcc is absent or a source does not compile
        if not os.path.exists(_SO):
            raise CblError("libcbl_amd.so is missing: run `python -m contrastboundary_amd.build`")
        override = os.environ.get("CBL_AMD_LIB")                     # another build of the same library (kernel experiments)
        _lib = ctypes.CDLL(override or _SO)
        if not hasattr(_lib, "cbl_version"):
            raise CblError(f"{override or _SO} is not a build of this library (no cbl_version)")
        _lib.cbl_version.restype = ctypes.c_char_p
        if override:
            # an override must be the SAME ABI: the version string of the in-tree build (csrc/version.hip) and every declared symbol (checked below)
            mine = ctypes.CDLL(_SO)
            mine.cbl_version.restype = ctypes.c_char_p
            if _lib.cbl_version() != mine.cbl_version():
                raise CblError(f"CBL_AMD_LIB={override}: version {_lib.cbl_version()!r} does not match the in-tree library's {mine.cbl_version()!r}")
        for name in declared_symbols():
            fn = getattr(_lib, name, None)
            if fn is None:
                raise CblError(f"libcbl_amd.so does not export {name} (declared in include/cbl_amd.h)")
            if name.endswith("_bytes"):
                fn.restype = ctypes.c_size_t
    return _lib


def check(code, what):
    if code != 0:
        raise CblError(f"{what} failed with code {code} "
                       f"({'bad argument' if code == -1 else 'workspace too small' if code == -2 else 'unsupported' if code == -3 else 'hipError_t'})")


def ptr(t):
    """device pointer of a torch tensor (or None) as c_void_p"""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream_of(t):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
