"""host issue time vs device time of the two halves of the ConvNet step (python tools/convnet_host_split.py):
the pyramid (data-dependent sizes: host waits inside) and the model part (AdaptiveWeight, labels, CBL) on a fixed pyramid"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrastboundary_amd import convnet_path as CP  # noqa: E402

scene = CP.ConvNetScene(200000, seed=0, b=1)
stage_list = CP.stages(scene, backward=True)
pyr_stage, model_stages = stage_list[0], stage_list[1:]
state = {}
for _ in range(10):
    CP.run_once(scene, state, stage_list=stage_list)
torch.cuda.synchronize()
R = 20
t = time.perf_counter()
for _ in range(R):
    pyr_stage[1](state)
torch.cuda.synchronize()
print("pyramid alone: wall ms %.3f" % ((time.perf_counter() - t) / R * 1e3))
t = time.perf_counter()
for _ in range(R):
    for _, fn, _, _ in model_stages:
        fn(state)
issue = time.perf_counter() - t
torch.cuda.synchronize()
print("model part: host issue ms %.3f, wall ms %.3f" % (issue / R * 1e3, (time.perf_counter() - t) / R * 1e3))

# pyramid of the NEXT scene on a thread + stream of its own beside the model part of this one (pure Python threads)
side = torch.cuda.Stream()
box = {}


def build():
    with torch.cuda.stream(side):
        s2 = {}
        pyr_stage[1](s2)
        box["pyr"] = s2["pyr"]
        box["ev"] = side.record_event()


for trial in range(2):
    build()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(R):
        torch.cuda.current_stream().wait_event(box["ev"])
        state["pyr"] = box["pyr"]
        th = threading.Thread(target=build)
        th.start()
        for _, fn, _, _ in model_stages:
            fn(state)
        th.join()
    torch.cuda.synchronize()
    print("overlapped (python thread): wall ms %.3f" % ((time.perf_counter() - t) / R * 1e3))

# single host thread, two streams: the model part of scene i is issued first (stream B), then the pyramid of scene i+1 (stream A, host waits inside)
A, B = torch.cuda.Stream(), torch.cuda.Stream()
keep = []


def build_on(stream):
    with torch.cuda.stream(stream):
        s2 = {}
        pyr_stage[1](s2)
        return s2["pyr"], stream.record_event()


for trial in range(3):
    pyr, ev = build_on(A)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(R):
        with torch.cuda.stream(B):
            B.wait_event(ev)
            state["pyr"] = pyr
            for _, fn, _, _ in model_stages:
                fn(state)
        keep.append(pyr); keep[:] = keep[-3:]
        pyr, ev = build_on(A)
    torch.cuda.synchronize()
    print("one thread, two streams (model of i, then pyramid of i+1): wall ms %.3f" % ((time.perf_counter() - t) / R * 1e3))
