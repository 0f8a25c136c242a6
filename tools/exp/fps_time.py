import os, sys, torch, numpy as np
from contrastboundary_amd import pointops, synthetic as S
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n in (40960, 10240, 5000, 2560, 1500):
    xyz = torch.from_numpy(S.s_room(n, seed=0, scale=1.0 if n <= 40960 else 2.0)[0]).cuda()
    o = torch.tensor([n], dtype=torch.int32, device="cuda"); no = torch.tensor([n // 4], dtype=torch.int32, device="cuda")
    us = timeit(lambda: pointops.furthestsampling(xyz, o, no))
    print(f"variant {os.environ.get('CBL_FPS_VARIANT','0')}: n={n} -> {n//4}: {us:9.1f} us  ({us/(n//4):.3f} us/sample)")
