"""the bench step (hotpath.run_once) replayed from a hipGraph vs issued eagerly"""
import time, torch
from contrastboundary_amd import hotpath
sc = hotpath.Scene.synthetic(40960, 64, seed=0)
state = {}
for _ in range(5): hotpath.run_once(sc, 16, state)
torch.cuda.synchronize()
def wall(fn, reps=100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("eager  %.4f ms/step" % wall(lambda: hotpath.run_once(sc, 16, state)))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): hotpath.run_once(sc, 16, state)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    hotpath.run_once(sc, 16, state)
print("graph  %.4f ms/step" % wall(lambda: g.replay()))
