"""the bench step (hotpath.Schedule with the nested searches) replayed from a hipGraph vs issued eagerly"""
import time, torch
from contrastboundary_amd import hotpath
sc = hotpath.Scene.synthetic(40960, 64, seed=0)
st = hotpath.stages(sc, 16)
sched = hotpath.Schedule(st, overlap=False, hints=hotpath.search_hints(sc))
state = {}
for _ in range(20): sched.run(state)
torch.cuda.synchronize()
def wall(fn, reps=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); return (t1 - t0) / reps * 1e3, (time.perf_counter() - t0) / reps * 1e3
for rep in range(3):
    print("eager  host %.4f  completed %.4f ms/step" % wall(lambda: sched.run(state)))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
gstate = {}
with torch.cuda.stream(side):
    for _ in range(3): sched.run(gstate)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    sched.run(gstate)
for rep in range(3):
    print("graph  host %.4f  completed %.4f ms/step" % wall(lambda: g.replay()))
ref = hotpath.run_once(sc, 16, {})
g.replay(); torch.cuda.synchronize()
print("graph outputs equal eager:", torch.equal(gstate["idx"], ref["idx"]), torch.equal(gstate["grouped"], ref["grouped"]), torch.equal(gstate["kpconv"], ref["kpconv"]),
      float(gstate["cbl_loss"].detach()), float(ref["cbl_loss"].detach()))
