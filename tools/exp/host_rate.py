"""is the bench step limited by the host (python + launch calls) or by the device?  enqueue time vs completed time per step"""
import time, torch, sys
from contrastboundary_amd import hotpath
sc = hotpath.Scene.synthetic(40960, 64, seed=0)
st = hotpath.stages(sc, 16)
for overlap in (True, False, True):
    sched = hotpath.Schedule(st, overlap=overlap)
    state = {}
    for _ in range(50): sched.run(state)
    torch.cuda.synchronize()
    N = 300
    t0 = time.perf_counter()
    for _ in range(N): sched.run(state)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("overlap=%s: host enqueue %.1f us/step, completed %.1f us/step" % (overlap, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
# per-stage host cost
state = {}
hotpath.run_once(sc, 16, state); torch.cuda.synchronize()
for name, fn, _, _ in st[:-1]:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fn(state)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("  %-22s host %.1f us per call" % (name, (t1 - t0) / 200 * 1e6))
