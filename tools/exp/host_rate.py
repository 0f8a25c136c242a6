"""is the bench step limited by the host (python + launch calls) or by the device?  enqueue time vs completed time per step"""
import time, torch
from contrastboundary_amd import hotpath
sc = hotpath.Scene.synthetic(40960, 64, seed=0)
st = hotpath.stages(sc, 16)
for name, kw in (("in order", dict(overlap=False)), ("nested searches", dict(overlap=False, hints=hotpath.search_hints(sc))), ("two streams", dict(overlap=True)), ("nested searches", dict(overlap=False, hints=hotpath.search_hints(sc)))):
    sched = hotpath.Schedule(st, **kw)
    state = {}
    for _ in range(50): sched.run(state)
    torch.cuda.synchronize()
    N = 300
    t0 = time.perf_counter()
    for _ in range(N): sched.run(state)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-16s host enqueue %.1f us/step, completed %.1f us/step" % (name, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
    del sched, state
    torch.cuda.synchronize()
