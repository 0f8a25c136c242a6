#!/usr/bin/env python3
"""KPConv forward / backward (csrc/local_aggregation.hip, kpconv_backward.hip) alone at the headline shape (S-room 40960, K = 16, C = 64, 15 kernel points, cell order):
20 replays of a hipGraph of the call between two HIP events / 20 -> one JSON line.  CBL_KPCONV_FWD=0|1|2 picks the forward kernel (A/B inside one gpurun call)."""
import json, os, sys
import torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from contrastboundary_amd import hotpath, pointops, local_aggregation

def graph_us(fn, reps=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        keep = fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = []
    for _ in range(3):
        with torch.cuda.stream(s):
            g.replay(); a.record()
            for _ in range(reps):
                g.replay()
            b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / reps * 1e3)
    return round(sorted(out)[1], 2)

n, c, k = 40960, 64, 16
sc = hotpath.Scene.synthetic(n, c, 0)
with pointops.neighbor_cache():
    idx, _ = pointops.knnquery_raw(k, sc.xyz, sc.xyz, sc.offset, sc.offset)
    fwd = graph_us(lambda: local_aggregation.kpconv(sc.xyz, sc.xyz, idx, sc.feat, sc.kernel_points, sc.kernel_weights, 0.12))
    out = local_aggregation.kpconv(sc.xyz, sc.xyz, idx, sc.feat, sc.kernel_points, sc.kernel_weights, 0.12)
    torch.cuda.synchronize()
    print(json.dumps({"variant": os.environ.get("CBL_KPCONV_FWD", "default"), "kpconv_fwd_us": fwd, "checksum": float(out.double().abs().sum().item())}))
