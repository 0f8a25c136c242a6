"""how many queries of a search go to the exact replay (reads the worklist counter out of the search workspace), shuffled vs Morton-ordered scene"""
import numpy as np, torch
from contrastboundary_amd import pointops, synthetic as S
def spread(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x030000ff; v = (v | (v << 8)) & 0x0300f00f; v = (v | (v << 4)) & 0x030c30c3; v = (v | (v << 2)) & 0x09249249
    return v
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
n = 40960
base = S.s_room(n, seed=0)[0]
q = np.clip(((base - base.min(0)) * (1023.99 / (base.max(0) - base.min(0)).max())).astype(np.int64), 0, 1023)
perm = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")
radial = np.argsort(((base - base.mean(0)) ** 2).sum(1), kind="stable")
off = torch.tensor([n], dtype=torch.int32, device="cuda")
for name, order in (("shuffled", np.arange(n)), ("morton", perm), ("radial", radial)):
    xyz = torch.from_numpy(np.ascontiguousarray(base[order])).cuda()
    for K, algo in ((16, "reference"), (16, "set"), (36, "reference"), (8, "reference")):
        idx, d2 = pointops.knnquery_raw(K, xyz, xyz, off, off, algo=algo)
        torch.cuda.synchronize()
        ws = pointops._workspace(1, xyz.device)
        cnt = ws[256:260].view(torch.int32).item()
        t = timeit(lambda: pointops.knnquery_raw(K, xyz, xyz, off, off, algo=algo))
        print("%-9s K=%2d %-9s worklist %4d  %7.1f us" % (name, K, algo, cnt, t))
