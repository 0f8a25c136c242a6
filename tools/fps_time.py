#!/usr/bin/env python3
"""furthest point sampling of the Point Transformer's four down-sampling stages (40960 -> 10240 -> 2560 -> 640 -> 160), HIP-event medians"""
import json, sys
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from contrastboundary_amd import pointops, synthetic as S
xyz, _ = S.s_room(40960, 0)
p = torch.from_numpy(xyz).cuda(); o = torch.tensor([40960], dtype=torch.int32, device="cuda")
out = {}
for m in (10240, 2560, 640, 160):
    no = torch.tensor([m], dtype=torch.int32, device="cuda")
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); idx = pointops.furthestsampling(p, o, no); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    out["%d->%d" % (p.shape[0], m)] = round(float(np.median(ts[2:])), 3)
    p = p[idx.long()].contiguous(); o = no
# four scenes in one batch (4 workgroups)
xs = np.concatenate([S.s_room(40960, i)[0] for i in range(4)])
p = torch.from_numpy(xs).cuda(); o = torch.tensor([40960 * (i + 1) for i in range(4)], dtype=torch.int32, device="cuda"); no = o // 4
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); pointops.furthestsampling(p, o, no); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
out["4 x 40960->10240"] = round(float(np.median(ts[1:])), 3)
# the network's four stages as its TransitionDown blocks run them (pointops.fps_downsample): with the prefix certificates of cbl_furthestsampling_chain, and with
# every stage sampled
xyz, _ = S.s_room(40960, 0)
for chain in (True, False):
    pointops.fps_prefix_chain = chain
    ts = []
    for _ in range(6):
        p = torch.from_numpy(xyz).cuda(); o = torch.tensor([40960], dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4):
            p, o, _ = pointops.fps_downsample(p, o, 4)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    out["chain 40960->160, %s" % ("prefix certificates" if chain else "every stage sampled")] = round(float(np.median(ts[2:])), 3)
pointops.fps_prefix_chain = True
print(json.dumps(out))
