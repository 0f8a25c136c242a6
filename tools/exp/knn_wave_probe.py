"""K > 16 grid KNN: time the wave-per-query kernel (set CBL_KNN_WAVE_MIN_K=100 to get the group kernels)."""
import os, sys, torch
from contrastboundary_amd import pointops, hotpath, synthetic as S
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
sc = hotpath.Scene.synthetic(40960, 64, seed=0)
import numpy as np
xu = torch.from_numpy(S.s_uniform(40960, seed=1)[0] if isinstance(S.s_uniform(40960, seed=1), tuple) else S.s_uniform(40960, seed=1)).cuda().float()
for name, xyz in (("room", sc.xyz), ("uniform", xu)):
    off = torch.tensor([xyz.shape[0]], dtype=torch.int32, device="cuda")
    for K in (20, 24, 32, 36, 48, 64):
        for algo in ("set", "auto"):
            us = timeit(lambda: pointops.knnquery_raw(K, xyz, xyz, off, off, algo=algo))
            print(f"{name:8s} K={K:3d} {algo:5s} {us:8.1f} us  (CBL_KNN_WAVE_MIN_K={os.environ.get('CBL_KNN_WAVE_MIN_K','17')})")
