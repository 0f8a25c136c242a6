#!/usr/bin/env python3
"""dense.batch_norm (csrc/bn_rows.hip) forward / backward at the network's (rows, C) shapes, with and without the residual tail: HIP-event medians in us"""
import json, sys
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from contrastboundary_amd import dense

def med(fn, reps=30):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return round(float(np.median(ts[5:])), 1)

out = {}
for rows, C in ((163840, 32), (40960, 32), (40960, 64), (10240, 64), (2560, 128), (640, 256)):
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    x = torch.randn(rows, C, device="cuda", requires_grad=True); r = torch.randn(rows, C, device="cuda", requires_grad=True); g = torch.randn(rows, C, device="cuda")
    for tag, res in (("bn_relu", None), ("bn_res_relu", r)):
        y = dense.batch_norm(x, bn, relu=True, residual=res)
        f = med(lambda: dense.batch_norm(x, bn, relu=True, residual=res))
        bw = med(lambda: torch.autograd.grad(y, [x] + ([r] if res is not None else []) + [bn.weight, bn.bias], g, retain_graph=True))
        out["%dx%d %s" % (rows, C, tag)] = {"fwd_us": f, "bwd_us": bw}
print(json.dumps(out))
