#!/usr/bin/env python3
"""Which python lines of the mirrors launch the library's own small kernels (reductions, copies, fills, elementwise adds ...) in a training step?
One eager step of the Point Transformer + CBL network under torch.profiler with stacks; device kernels that are NOT this library's (names without
"anonymous namespace" / our prefixes) are grouped by the innermost frame inside contrastboundary_amd/ that issued them.
    python tools/model_glue_profile.py [--scenes 1] [--top 40]      -> table on stdout"""
import argparse, collections, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from contrastboundary_amd import pointtransformer_seg as M, synthetic as S, neighbor_state

ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=1); ap.add_argument("--top", type=int, default=45); ap.add_argument("--n", type=int, default=40960)
a = ap.parse_args()
torch.backends.cuda.preferred_blas_library("cublas")
cfg = M.Config({"base_fdim": 32, "nsample": [36, 24, 24, 24, 24], "nstride": [4, 4, 4, 4], "ignore_label": 255, "voxel_size": 0.04,
                "contrast": {"stage": "Ua", "contrast": "softnn", "ftype": "latent", "sample": "label", "pos": "cnt", "dist": "l2", "temperature": 1, "weight": "w.1"},
                "multi": {"stage": "Ua", "ftype": "latent", "combine": "concat"}})
torch.manual_seed(0)
model = M.pointtransformer_seg_repro(c=6, k=13, config=cfg).cuda().train()
crit = M.Loss(cfg)
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True)
xs, ls = zip(*[S.s_room(a.n, seed=i) for i in range(a.scenes)])
inputs = {"points": torch.from_numpy(np.concatenate(xs)).cuda(), "features": torch.rand(a.n * a.scenes, 3, device="cuda"),
          "offset": torch.tensor(np.cumsum([a.n] * a.scenes), dtype=torch.int32, device="cuda")}
target = torch.from_numpy(np.concatenate(ls)).cuda()

def step():
    opt.zero_grad(set_to_none=True)
    out, sl, loss, nc = M.forward_and_loss(model, crit, inputs, target)
    loss.sum().backward()
    neighbor_state.release_unowned_transposes()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ours = ("anonymous namespace", "cbl_", "grid_", "knn_", "nt_", "pt_", "fps", "Cijk_", "rocprim")
by_site = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
total_launch, total_glue = 0, 0
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CUDA and not getattr(ev, "kernels", None):
        continue
    kernels = getattr(ev, "kernels", None) or []
    for k in kernels:
        total_launch += 1
        if any(t in k.name for t in ours):
            continue
        total_glue += 1
        site = "?"
        for fr in (ev.stack or []):
            if "contrastboundary_amd/" in fr:
                site = fr.split("contrastboundary_amd/")[1]; break
        if site == "?" and ev.stack:
            site = "(outside) " + " < ".join(f.split("/")[-1] for f in ev.stack[:3])
        if site == "?":                                          # backward ops run on autograd's thread without a python stack: name the autograd node instead
            par = ev.cpu_parent
            while par is not None and site == "?":
                if "Backward" in par.name or par.name.startswith("autograd::"):
                    site = "(node) " + par.name
                par = par.cpu_parent
        e = by_site[(site, ev.name)]
        e[0] += 1; e[1] += k.duration; e[2][k.name[:60]] += 1
print("launches in the step: %d, of which library glue (not this package's kernels, not GEMMs): %d" % (total_launch, total_glue))
rows = sorted(by_site.items(), key=lambda kv: -kv[1][0])
for (site, op), (cnt, us, names) in rows[:a.top]:
    print("%4d  %8.1f us  %-34s %-60s %s" % (cnt, us, op[:34], site[:60], names.most_common(1)[0][0][:50]))
