"""Build libcbl_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m contrastboundary_amd.build [--force] [--verbose]

Objects are cached under contrastboundary_amd/lib/obj and rebuilt when the source or a header is newer.
The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
SO = os.path.join(LIBDIR, "libcbl_amd.so")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: every float expression keeps the CPU oracle's rounding (no silent FMA), see DESIGN.md
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-I" + INCLUDE, "-I" + CSRC]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return hs


# per-file flags.  local_aggregation.hip: MFMA results in VGPRs (gfx950 has one unified file): the KPConv kernels read every accumulator once per
# point in their epilogue, which through AGPRs is a v_accvgpr_read each (+ a v_accvgpr_write to zero it) on a kernel that is bound by its VALU issue slots
# (tried and dropped here: -fno-slp-vectorize, which raised the register count of the KPConv backward)
# (tried in round 3: -ffp-contract=fast for the float-output translation units — no kernel got faster (the hot multiply-adds that matter are written as
# fmaf where a kernel is VALU bound, local_aggregation.hip), and KPConv's 'closest' kernel-point choice flips on fused distances)
EXTRA_FLAGS = {"local_aggregation.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}

def _compile(src, obj, verbose):
    tmp = obj + ".tmp.%d" % os.getpid()
    cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if verbose and r.stderr.strip():
        print(r.stderr)
    os.replace(tmp, obj)


def is_stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(p) > t for p in sources() + headers())


def build(force=False, verbose=False):
    """compile what is out of date and link; safe to call from several processes at once (one rank per GPU): an exclusive file lock
    serialises them, objects and the library are written under temporary names and renamed into place"""
    import fcntl
    os.makedirs(OBJDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():
                return SO                                            # another process finished the build while this one waited
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    hdr_time = max([os.path.getmtime(h) for h in headers()] + [os.path.getmtime(__file__)])
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append((src, obj))
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for f in [ex.submit(_compile, s, o, verbose) for s, o in jobs]:
                f.result()
    if jobs or not os.path.exists(SO):
        tmp = SO + ".tmp.%d" % os.getpid()
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        os.replace(tmp, SO)
    else:
        os.utime(SO)                                                 # up to date: stop looking stale (e.g. a header touched without changes)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
