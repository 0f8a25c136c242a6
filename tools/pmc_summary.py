#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (two separate passes, tools/gpu_pmc.sh) -> profiles/rNN_pmc_traffic.json:
HBM-side bytes per launch of every kernel of `bench.py`, with the gfx950 correction of MI355X_MICROARCH.md's HBM section
(FETCH_SIZE is reported in KiB and counts 16-B-per-lane reads at half their size: x2; WRITE_SIZE in KiB, uncorrected)."""
import collections
import csv
import json
import re
import sys


def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    depth, out = 0, []
    for ch in n:                      # cut at the argument list: first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def main(fetch_csv, write_csv, out_json):
    acc = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for path, ctr in ((fetch_csv, "FETCH_SIZE"), (write_csv, "WRITE_SIZE")):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == ctr:
                acc[short(r["Kernel_Name"])][ctr].append(float(r["Counter_Value"]))
    doc = {"_meta": {"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline "
                                "(two separate passes, tools/gpu_pmc.sh; summarised by tools/pmc_summary.py)",
                     "workload": "N=40960 K=16 C=64 S-room seed 0, forward + backward of the block",
                     "kernels": "the kernel set of this commit's step (contrast_pairs / contrast_gather / nt_* / grouping_bwd_csr_rows / kpconv_bwd_csr)",
                     "correction": "FETCH_SIZE x2 (gfx950 rocprofv3 reports half the bytes of 16 B/lane reads, MI355X_MICROARCH.md HBM section); "
                                   "WRITE_SIZE uncorrected; counters count fabric-side requests, Infinity-Cache hits included"}}
    for k, v in acc.items():
        f = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)
        w = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
        doc[k] = {"FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w, "launches": max(len(v["FETCH_SIZE"]), len(v["WRITE_SIZE"])),
                  "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0, "hbm_bytes_per_launch_uncorrected": (f + w) * 1024.0}
    json.dump(doc, open(out_json, "w"), indent=1)
    for k in sorted(acc, key=lambda k: -doc[k]["hbm_bytes_per_launch"])[:14]:
        print(f"{k[:60]:60s} {doc[k]['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch  ({doc[k]['launches']} launches)")


if __name__ == "__main__":
    main(*sys.argv[1:4])
