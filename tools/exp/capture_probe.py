"""which schedule / stage set crashes hipGraph capture"""
import sys, faulthandler, torch
faulthandler.enable()
from contrastboundary_amd import hotpath
def run(overlap, backward, mode):
    print("== overlap", overlap, "backward", backward, "mode", mode, flush=True)
    sc = hotpath.Scene.synthetic(16384, 32, seed=7)
    st = hotpath.stages(sc, 16, backward)
    sched = hotpath.Schedule(st, overlap=overlap, hints=hotpath.search_hints(sc))
    gstate = {}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            sched.run(gstate)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print("warm-up ok", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=mode):
        sched.run(gstate)
    print("captured", flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("replayed ok", flush=True)
which = sys.argv[1]
run(which[0] == "o", which[1] == "b", "thread_local" if which[2] == "t" else "global")
