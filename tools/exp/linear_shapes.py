import collections, torch, numpy as np
import torch.nn.functional as F
from contrastboundary_amd import dense, pointtransformer_seg as M, synthetic as S
shapes = collections.Counter()
orig = F.linear
def spy(x, w, b=None):
    shapes[(x.numel() // w.shape[1], w.shape[1], w.shape[0])] += 1
    return orig(x, w, b)
F.linear = spy; dense.F.linear = spy
cfg = M.Config({"base_fdim": 32, "nsample": [36, 24, 24, 24, 24], "nstride": [4, 4, 4, 4], "ignore_label": 255, "contrast": {"stage": "Ua", "contrast": "softnn", "ftype": "latent", "sample": "label", "pos": "cnt", "dist": "l2", "temperature": 1, "weight": "w.1"}, "multi": {"stage": "Ua", "ftype": "latent", "combine": "concat"}})
model = M.pointtransformer_seg_repro(c=6, k=13, config=cfg).cuda().train(); crit = M.Loss(cfg)
xyz, lab = S.s_room(40960, 0)
inputs = {"points": torch.from_numpy(xyz).cuda(), "features": torch.rand(40960, 3, device="cuda"), "offset": torch.tensor([40960], dtype=torch.int32, device="cuda")}
out, sl, loss, nc = M.forward_and_loss(model, crit, inputs, torch.from_numpy(lab).cuda())
for k, v in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2]):
    print(k, v, f"{2*k[0]*k[1]*k[2]/1e9:.3f} GFLOP")
