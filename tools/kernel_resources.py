"""Static resource table of every kernel in the built library: registers, LDS, scratch and spill counts as the compiler recorded them in the gfx950 code
objects' metadata (no GPU needed).

    python tools/kernel_resources.py [--all]      # kernels with scratch or spills (or, with --all, every kernel), widest first

Used by tests/test_kernel_resources.py: none of our kernels may spill vector registers or touch scratch memory (a spilled VGPR is an HBM round trip per
use; rocprim's radix sort keeps 80 bytes of scratch of its own)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "contrastboundary_amd", "lib", "obj")
LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = ("name", "vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count",
        "max_flat_workgroup_size")


def kernels():
    """-> list of dicts (file, name (demangled), and the integer fields of KEYS) for every kernel of every object of the built library"""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(OBJ, "*.o"))):
            base = os.path.splitext(os.path.basename(obj))[0]
            fat, co = os.path.join(tmp, base + ".fatbin"), os.path.join(tmp, base + ".co")
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
            if not os.path.exists(fat) or os.path.getsize(fat) == 0:
                continue                                              # a file without kernels (version.hip)
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                "--input=" + fat, "--output=" + co], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co):
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            # the metadata lists kernels as YAML maps whose keys are sorted: every map starts at "- .agpr_count" (or "- .args")
            for block in re.split(r"\n\s+- \.(?=agpr_count|args)", notes)[1:]:
                cur = {}
                for key in KEYS:
                    m = re.search(r"^\s*\.?%s:\s+(\S+)\s*$" % key, ("." + block) if block.startswith(key) else block, re.M)
                    if m:
                        cur[key] = m.group(1)
                if "name" not in cur or "vgpr_count" not in cur:
                    continue
                k = {key: (int(v) if key != "name" else v) for key, v in cur.items()}
                k["file"] = base
                out.append(k)
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in out), capture_output=True, text=True).stdout.splitlines()
    for k, n in zip(out, names):
        k["name"] = n
    return out


def ours(k):
    return "rocprim" not in k["name"]


if __name__ == "__main__":
    ks = kernels()
    show = ks if "--all" in sys.argv else [k for k in ks if k.get("private_segment_fixed_size", 0) or k.get("vgpr_spill_count", 0) or k.get("sgpr_spill_count", 0)]
    print("%d kernels in %d files; %d of ours with scratch or VGPR spills" % (
        len(ks), len({k["file"] for k in ks}), sum(1 for k in ks if ours(k) and (k.get("private_segment_fixed_size", 0) or k.get("vgpr_spill_count", 0)))))
    print("%-18s %5s %5s %5s %7s %7s %6s %6s  %s" % ("file", "vgpr", "agpr", "sgpr", "lds", "scratch", "vspill", "sspill", "kernel"))
    for k in sorted(show, key=lambda k: (-k["vgpr_count"], k["name"])):
        print("%-18s %5d %5d %5d %7d %7d %6d %6d  %s" % (k["file"], k["vgpr_count"], k.get("agpr_count", 0), k.get("sgpr_count", 0), k.get("group_segment_fixed_size", 0),
                                                       k.get("private_segment_fixed_size", 0), k.get("vgpr_spill_count", 0), k.get("sgpr_spill_count", 0), k["name"][:150]))
