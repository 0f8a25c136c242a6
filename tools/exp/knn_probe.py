"""GPU experiment: where does the KNN time go? (worklist size, replay latency, exact kernel vs m)"""
import time, numpy as np, torch, ctypes
from contrastboundary_amd import pointops, _lib, hotpath

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps): fn()
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps * 1e3   # us

sc = hotpath.Scene.synthetic(40960, 64, seed=0)
xyz, off = sc.xyz, sc.offset
n = xyz.shape[0]
for k in (16, 36):
    idx, d2 = pointops.knnquery_raw(k, xyz, xyz, off, off)
    torch.cuda.synchronize()
    ws = list(pointops._ws_cache.values())[0]
    # counters live right after the CblGrid array (48 B * b, 256-aligned)
    cnt = ws[256:260].view(torch.int32).item()
    print(f"K={k}: worklist (replayed) queries = {cnt}; auto = {timeit(lambda: pointops.knnquery_raw(k, xyz, xyz, off, off)):.1f} us")
for m in (1, 2, 8, 64, 1024, 40960):
    q = xyz[:m].contiguous(); qo = torch.tensor([m], dtype=torch.int32, device="cuda")
    print(f"exact kernel, m={m:6d} queries x n={n}: {timeit(lambda: pointops.knnquery_raw(16, xyz, q, off, qo, algo='exact'), 5):.1f} us")
print("--- single query latency vs n (scan vs insert cost)")
for nn in (1024, 4096, 16384, 40960):
    x = xyz[:nn].contiguous(); o = torch.tensor([nn], dtype=torch.int32, device="cuda")
    q = x[:1].contiguous(); qo = torch.tensor([1], dtype=torch.int32, device="cuda")
    print(f"m=1, n={nn:6d}: {timeit(lambda: pointops.knnquery_raw(16, x, q, o, qo, algo='exact'), 10):.1f} us")
