import sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench
from contrastboundary_amd import hotpath
torch.backends.cuda.preferred_blas_library("cublas")
mode = sys.argv[1]
import os
CM = os.environ.get("CM", "thread_local")
_orig = torch.cuda.graph.__init__
def _init(self, *a, **k):
    if "capture_error_mode" in k: k["capture_error_mode"] = CM
    _orig(self, *a, **k)
torch.cuda.graph.__init__ = _init
args = bench.parse(["--block", "pt"])
scene = hotpath.Scene.synthetic(40960, 64, seed=0, b=1)
if mode == "eager":
    step = bench.Step(scene, 16, True, args, overlap=True, pipeline=False)
    bench.settle(step, 0.1); step(); torch.cuda.synchronize(); print("eager ok", flush=True)
elif mode == "graph":
    step = bench.Step(scene, 16, True, args, overlap=True, pipeline=False)
    bench.settle(step, 0.1); step.capture(); step(); torch.cuda.synchronize(); print("single graph ok", step.note, flush=True)
elif mode == "graph_noov":
    step = bench.Step(scene, 16, True, args, overlap=False, pipeline=False)
    bench.settle(step, 0.1); step.capture(); step(); torch.cuda.synchronize(); print("single graph no overlap ok", flush=True)
elif mode == "pipe_fwd":
    step = bench.Step(scene, 16, False, args, overlap=True, pipeline=True)
    bench.settle(step, 0.1); step.capture(); step(); torch.cuda.synchronize(); print("pipeline forward-only ok", flush=True)
elif mode == "pipe":
    step = bench.Step(scene, 16, True, args, overlap=True, pipeline=True)
    bench.settle(step, 0.1); step.capture(); step(); torch.cuda.synchronize(); print("pipeline ok", flush=True)
