#!/usr/bin/env python3
"""Static instruction mix of a kernel's loops, from the gfx950 assembly hipcc emits — no GPU needed.

    python tools/isa_loop_mix.py contrastboundary_amd/csrc/pt_layer.hip 'pt_w2_bwd_kernel<64, 16, false>' [more kernel-name substrings ...]

Compiles the source with the library's flags (`-S --cuda-device-only`), finds every backward branch of each named kernel (a loop = the
lines between a label and the last branch back to it) and prints, per loop and for the whole kernel, how many vector-ALU, matrix, scalar,
LDS, vector-memory and wait instructions it holds.  The passes of pt_layer.hip / local_aggregation.hip are bound by their waves' vector
issue slots, so the VALU count of the main loop is the number a change must move before a GPU minute is spent on it."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def assembly(src):
    from contrastboundary_amd import build as B
    out = "/tmp/isa_%s.s" % os.path.basename(src).replace(".", "_")
    cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(os.path.basename(src), []) + ["-S", "--cuda-device-only", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stderr)
    return open(out).read().splitlines()


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return [s.replace("(anonymous namespace)::", "").replace("void ", "") for s in r.stdout.splitlines()]


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load") or op.startswith("scratch_load"):
        return "vmem_ld"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store") or op.startswith("scratch_store"):
        return "vmem_st"
    if op.startswith("global_atomic") or op.startswith("buffer_atomic"):
        return "atomic"
    return "other"


KEYS = ["valu", "mfma", "acc_mov", "salu", "lds", "vmem_ld", "vmem_st", "atomic", "wait", "barrier", "branch", "nop", "other"]


def mix(lines):
    c = dict.fromkeys(KEYS, 0)
    for l in lines:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        c[classify(t.split()[0])] += 1
    return c


def fmt(c):
    return "  ".join("%s %d" % (k, c[k]) for k in KEYS if c[k])


def main():
    src, wanted = sys.argv[1], sys.argv[2:]
    text = assembly(os.path.join(ROOT, src) if not os.path.isabs(src) else src)
    starts = [(i, m.group(1)) for i, l in enumerate(text) for m in [re.match(r"^(_Z\S+):", l)] if m]
    names = demangle([s for _, s in starts])
    for (i, _), name in zip(starts, names):
        if wanted and not any(w in name for w in wanted):
            continue
        end = next((j for j in range(i + 1, len(text)) if text[j].strip().startswith("s_endpgm")), len(text))
        body = text[i:end + 1]
        labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\S+):", l)] if m}
        loops = {}
        for k, l in enumerate(body):
            m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\S+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                loops[m.group(1)] = max(loops.get(m.group(1), 0), k)
        print("== %s" % name.split("(")[0])
        print("   whole kernel : %s" % fmt(mix(body)))
        for lab, k in sorted(loops.items(), key=lambda kv: labels[kv[0]]):
            c = mix(body[labels[lab]:k + 1])
            if c["valu"] + c["mfma"] >= 16:
                print("   loop %-12s (%5d lines): %s" % (lab, k - labels[lab], fmt(c)))


if __name__ == "__main__":
    main()
