#!/usr/bin/env python3
"""Print VGPR/SGPR/spill/LDS/occupancy per kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys

import os
ROOT = os.path.abspath(__file__).rsplit("/tools/", 1)[0]
sys.path.insert(0, ROOT)
from contrastboundary_amd import build as B                         # the library's own flags, per-file extras included

for src in sys.argv[1:]:
    r = subprocess.run([B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(os.path.basename(src), []) +
                       ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            if cur:
                print(cur)
            name = t.split(":", 1)[1].strip()
            d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(anonymous namespace\)::", "", d).split("(")[0][:60].ljust(60)
        else:
            k, v = [s.strip() for s in t.split(":", 1)]
            short = {"VGPRs": "V", "AGPRs": "A", "TotalSGPRs": "S", "ScratchSize [bytes/lane]": "scr", "Occupancy [waves/SIMD]": "occ",
                     "LDS Size [bytes/block]": "lds", "VGPR Spill": "vspill", "SGPR Spill": "sspill"}.get(k)
            if short:
                cur += f" {short}={v}"
    if cur:
        print(cur)
