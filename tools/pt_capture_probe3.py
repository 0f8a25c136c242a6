import sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from contrastboundary_amd import hotpath, pointops
torch.backends.cuda.preferred_blas_library("cublas")
which = sys.argv[1].split(",")
use_cache = len(sys.argv) > 2 and sys.argv[2] == "cache"
scene = hotpath.Scene.synthetic(40960, 64, seed=0, b=1)
stages = {s[0]: s[1] for s in hotpath.stages_pt(scene, 16, True)}
def body():
    st = {}
    if use_cache:
        with pointops.neighbor_cache() as nc:
            for xyz, ns, algo in hotpath.search_hints(scene): nc.hint(xyz, ns, algo)
            for nm in which: stages[nm](st)
    else:
        for nm in which: stages[nm](st)
    return st
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
g.replay(); torch.cuda.synchronize()
print(which, use_cache, "ok", flush=True)
