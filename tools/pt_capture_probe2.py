import sys, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from contrastboundary_amd import hotpath, pointops
torch.backends.cuda.preferred_blas_library("cublas")
mode = sys.argv[1]
scene = hotpath.Scene.synthetic(40960, 64, seed=0, b=1)
layer = hotpath.pt_layer(scene)
idx, _ = pointops.knnquery_raw(16, scene.xyz, scene.xyz, scene.offset, scene.offset)
up = scene.upstream(16)["grad_kpconv"]
params = list(layer.parameters())
def body():
    x = scene.feat.detach().requires_grad_(True)
    y = layer([scene.xyz, x, scene.offset], idx=idx)
    if mode == "x":
        return torch.autograd.grad(y, [x], up)
    if mode == "params":
        return torch.autograd.grad(y, [x] + params, up)
    if mode == "backward":
        for p in params: p.grad = None
        y.backward(up); return x.grad
    if mode == "unfused":
        return torch.autograd.grad(y, [x] + params, up)
if mode == "unfused":
    layer.fused = False
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
g.replay(); torch.cuda.synchronize()
print(mode, "ok", flush=True)
