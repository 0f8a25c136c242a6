#!/usr/bin/env python3
"""How much of the neighbour-row kernels' time is the gather itself?  The same launches with neighbour tables that need fewer DISTINCT rows per point:
   real     the K = 16 (K = 36 for the CBL pair kernel) table of the S-room scene, cell order
   quarter  every row's columns cycle over its first K/4 neighbours (a quarter of the distinct rows per point, same arithmetic)
   self     every column is the point itself (one row per point: the arithmetic alone, rows from L1)
-> one JSON line, us per launch (hipGraph replays between two HIP events).  python tools/gather_bound_probe.py"""
import json, sys
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from contrastboundary_amd import hotpath, pointops, local_aggregation, heads

def graph_us(fn, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        keep = fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        g.replay(); a.record()
        for _ in range(reps):
            g.replay()
        b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps * 1e3, 1)

n, c, k = 40960, 64, 16
sc = hotpath.Scene.synthetic(n, c, 0)
out = {}
with pointops.neighbor_cache() as nc:
    nc.hint(sc.xyz, 36, "set")
    idx, _ = pointops.knnquery_raw(k, sc.xyz, sc.xyz, sc.offset, sc.offset)
    widx, _ = pointops.knnquery_raw(36, sc.xyz, sc.xyz, sc.offset, sc.offset, algo="set")
    order = pointops.spatial_order(idx)
    def variants(t):
        kk = t.shape[1]
        q = max(kk // 4, 1)
        quarter = t[:, torch.arange(kk, device=t.device) % q].contiguous()
        self_ = torch.arange(n, dtype=torch.int32, device=t.device)[:, None].expand(n, kk).contiguous()
        return {"real": t, "quarter": quarter, "self": self_}
    layer = hotpath.pt_layer(sc)
    for name, t in variants(idx).items():
        pointops.neighbor_state._order_alias(t, sc.xyz)             # the same processing order for every variant
        r = {}
        r["queryandgroup"] = graph_us(lambda: pointops.queryandgroup(k, sc.xyz, sc.xyz, sc.feat, t, sc.offset, sc.offset, use_xyz=True))
        r["kpconv_fwd"] = graph_us(lambda: local_aggregation.kpconv(sc.xyz, sc.xyz, t, sc.feat, sc.kernel_points, sc.kernel_weights, 0.12))
        with torch.no_grad():
            r["pt_layer_fwd"] = graph_us(lambda: layer([sc.xyz, sc.feat, sc.offset], idx=t))
        out[name] = r
    for name, t in variants(widx).items():
        pointops.neighbor_state._order_alias(t, sc.xyz)
        with torch.no_grad():
            out[name]["cbl_pairs_fwd_nograd"] = graph_us(lambda: heads.point_contrast(sc.latent, sc.labels, t, 1.0, 0.1))
print(json.dumps(out))
