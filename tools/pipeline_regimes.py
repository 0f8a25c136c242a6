"""does the pipelined step's rate change over time?  python tools/pipeline_regimes.py  (per-block wall time of consecutive blocks of steps, no sync in between
except the closing one of each block)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from contrastboundary_amd import hotpath  # noqa: E402

args = bench.parse([])
scene = hotpath.Scene.synthetic(40960, 64, seed=0, b=1)
step = bench.make_step(scene, 16, True, args, overlap=True, pipeline=True)
print("tuning:", getattr(step.pipe, "tuning", None))
for blk in (25, 50, 100, 200, 400, 100, 50, 25):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(blk):
        step()
    ti = time.perf_counter() - t0
    step.join(); torch.cuda.synchronize()
    print("block of %4d steps: %.4f ms per step (host issue %.4f)" % (blk, (time.perf_counter() - t0) / blk * 1e3, ti / blk * 1e3))
