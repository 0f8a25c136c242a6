import time, torch, numpy as np
from contrastboundary_amd import pointtransformer_seg as M, synthetic as S
torch.backends.cuda.preferred_blas_library("cublas")
cfg = M.Config({"base_fdim": 32, "nsample": [36, 24, 24, 24, 24], "nstride": [4, 4, 4, 4], "ignore_label": 255, "contrast": {"stage": "Ua", "contrast": "softnn", "ftype": "latent", "sample": "label", "pos": "cnt", "dist": "l2", "temperature": 1, "weight": "w.1"}, "multi": {"stage": "Ua", "ftype": "latent", "combine": "concat"}})
torch.manual_seed(0)
model = M.pointtransformer_seg_repro(c=6, k=13, config=cfg).cuda().train(); crit = M.Loss(cfg)
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
xyz, lab = S.s_room(40960, 0)
inputs = {"points": torch.from_numpy(xyz).cuda(), "features": torch.rand(40960, 3, device="cuda"), "offset": torch.tensor([40960], dtype=torch.int32, device="cuda")}
target = torch.from_numpy(lab).cuda()
g = M.GraphedTrainStep(model, crit, opt, inputs, target)
def wall(fn, reps=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
s0 = g.sets[0]
print("replay only          %.2f ms" % wall(lambda: s0["graph"].replay()))
print("geometry refresh only %.2f ms" % wall(lambda: s0["geom"].refresh()))
g.stage(inputs, target)
def both():
    g.run(); g.stage(inputs, target)
print("run + stage           %.2f ms" % wall(both))
def staged_first():
    # stage for the other set BEFORE replaying this one
    g.run()
print("host time of replay call: ", end="")
t0 = time.perf_counter(); s0["graph"].replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); print("%.2f ms launch, %.2f total" % ((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
