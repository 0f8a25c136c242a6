"""grouping forward with index patterns of increasing locality: what bounds the gather kernel?"""
import numpy as np, torch
from contrastboundary_amd import pointops
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
n, K = 40960, 16
rng = np.random.default_rng(0)
for C in (64, 32, 128):
    f = torch.randn(n, C, device="cuda")
    pats = {"self (idx = row)": np.repeat(np.arange(n), K).reshape(n, K),
            "window +-64": (np.arange(n)[:, None] + rng.integers(-64, 64, (n, K))) % n,
            "window +-2048": (np.arange(n)[:, None] + rng.integers(-2048, 2048, (n, K))) % n,
            "random": rng.integers(0, n, (n, K))}
    for name, idx in pats.items():
        it = torch.from_numpy(idx.astype(np.int32)).cuda()
        us = timeit(lambda: pointops.grouping(f, it))
        by = 4 * n * K + 4 * n * C + 4 * n * K * C
        print(f"C={C:4d} {name:18s} {us:7.1f} us  {by / us / 1e6:5.2f} TB/s algorithmic")
    out = torch.empty(n, K, C, device="cuda")
    us = timeit(lambda: out.fill_(1.0)); print(f"C={C:4d} fill of the output   {us:7.1f} us  {4*n*K*C/us/1e6:5.2f} TB/s")
