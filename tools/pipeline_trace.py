"""steady-state alignment of the pipelined step from a rocprofv3 kernel trace: python tools/pipeline_trace.py <kernel_trace.csv>
prints, for the last periods, when every big kernel starts / ends relative to the wave search of its period, and on which queue"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0].split("<")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
ev.sort()
waves = [e for e in ev if e[2] == "knn_grid_wave_kernel"]
if len(waves) < 30:
    print("too few steps", len(waves)); sys.exit(0)
periods = [(waves[i + 1][0] - waves[i][0]) / 1e3 for i in range(len(waves) - 25, len(waves) - 1)]
print("period us: mean %.1f min %.1f max %.1f" % (sum(periods) / len(periods), min(periods), max(periods)))
big = ("knn_grid_wave_kernel", "knn_replay_kernel", "query_group_lds_pipe", "kpconv_fwd_c64_kernel", "grouping_bwd_csr_rows_kernel", "kpconv_bwd_csr_kernel",
       "contrast_pairs_kernel", "contrast_gather_kernel", "nt_finish_kernel", "grid_count_kernel")
for k in range(len(waves) - 6, len(waves) - 3):
    t0, t1 = waves[k][0], waves[k + 1][0]
    line = []
    for s, e, n, q in ev:
        if t0 <= s < t1 and n in big:
            line.append("%s[q%s] %d-%d" % (n.replace("_kernel", "")[:14], q, (s - t0) // 1000, (e - t0) // 1000))
    print(" | ".join(line))
