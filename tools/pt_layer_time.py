"""Per-kernel and per-pass timing of the Point-Transformer layer (csrc/pt_layer.hip) at a stage shape, eager, HIP events:
    python tools/pt_layer_time.py [n K C] [--graph]      -> one JSON line (layer forward / backward: the fused path issued eagerly and as replayed hipGraphs ("graph"), and round 3's split kernels)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrastboundary_amd import blocks, pointops, synthetic as S  # noqa: E402


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def graphed(fn):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn()
    g.replay(); torch.cuda.synchronize()
    return g


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n, K, C = (int(v) for v in args[:3]) if len(args) >= 3 else (40960, 16, 64)
    xyz = torch.from_numpy(S.s_room(n, seed=0)[0]).cuda(); o = torch.tensor([n], dtype=torch.int32, device="cuda")
    torch.manual_seed(0)
    x = torch.randn(n, C, device="cuda"); g = torch.randn(n, C, device="cuda")
    idx, _ = pointops.knnquery(K, xyz, xyz, o, o)
    which = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--order=")]
    if which:
        # experiment: another processing order than the search's cell order — "morton:<cell edge in metres>" (Z-curve over cells of that edge, ids ascending
        # inside a cell), "hilbertish:<edge>" (Z-curve with the x runs of odd rows reversed), "index" (the rows as they are)
        from contrastboundary_amd import neighbor_state as NS
        kind, _, edge = which[0].partition(":")
        if kind == "index":
            NS.use_spatial_order = False
        else:
            q = ((xyz - xyz.min(0).values) / float(edge or 0.1)).long()
            code = torch.zeros(n, dtype=torch.long, device="cuda")
            for b in range(10):
                for d in range(3):
                    code |= ((q[:, d] >> b) & 1) << (3 * b + d)
            order = torch.argsort(code * n + torch.arange(n, device="cuda")).to(torch.int32).contiguous()
            cur = torch.cuda.current_stream()
            for t in (xyz, idx):
                NS._order_registry[NS._order_key(t)] = {cur.cuda_stream: (order, cur)}
        out_order = which[0]
    out = {"n": n, "K": K, "C": C}
    if which:
        out["order"] = out_order
    for mode in ((True, "ops") if C > 64 else (True, "split")):
        layer = blocks.PointTransformerLayer(C, C, 8, K).cuda().train(); layer.fused = mode
        params = list(layer.parameters())
        state = {}

        def fwd():
            state["x"] = x.detach().requires_grad_(True)
            state["y"] = layer([xyz, state["x"], o], idx=idx)

        def bwd():
            torch.autograd.grad(state["y"], [state["x"]] + params, g, retain_graph=True)

        def both():
            fwd(); torch.autograd.grad(state["y"], [state["x"]] + params, g)

        if mode is True and "--graph" in sys.argv:
            # the two passes as replayed hipGraphs, BEFORE anything of this layer runs on the default stream (a parameter's gradient accumulator remembers the stream
            # it was created on; a capture that meets one from the default stream crashes on ROCm 7.2): device time with the launch gaps a graph leaves, without
            # the interpreter's issue time — the eager figures below are bounded by it at the small shapes (~25 launches of 5 - 25 us each)
            gf = graphed(fwd); gb = graphed(bwd)
            out["graph"] = {"fwd_us": round(timed(gf.replay), 1), "bwd_us": round(timed(gb.replay), 1)}
            out["graph"]["fwd_bwd_us"] = round(out["graph"]["fwd_us"] + out["graph"]["bwd_us"], 1)
            del gf, gb
            print("graph", out["graph"], file=sys.stderr, flush=True)
        fwd(); torch.cuda.synchronize(); print('fwd ok', mode, file=sys.stderr, flush=True)
        bwd(); torch.cuda.synchronize(); print('bwd ok', mode, file=sys.stderr, flush=True)
        tag = "new" if mode is True else str(mode)
        out[tag] = {"fwd_us": round(timed(fwd), 1), "bwd_us": round(timed(bwd), 1), "fwd_bwd_us": round(timed(both), 1)}
        print(tag, out[tag], file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
