"""where the host time of the eagerly issued ConvNet step goes: cProfile over 20 steps (python tools/convnet_host_profile.py)"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrastboundary_amd import convnet_path as CP  # noqa: E402

scene = CP.ConvNetScene(200000, seed=0, b=1)
stage_list = CP.stages(scene, backward=True)
state = {}
for _ in range(10):
    CP.run_once(scene, state, stage_list=stage_list)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    CP.run_once(scene, state, stage_list=stage_list)
issue = time.perf_counter() - t
torch.cuda.synchronize()
print("issue ms/step %.3f  wall ms/step %.3f" % (issue / 20 * 1e3, (time.perf_counter() - t) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    CP.run_once(scene, state, stage_list=stage_list)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
