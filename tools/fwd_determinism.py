"""is the network's forward bit-identical from run to run?  (python tools/fwd_determinism.py)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_model import build, CASES  # noqa: E402
from contrastboundary_amd import pointops, pt_layer  # noqa: E402

M, model, crit, g = build(CASES[0])
model = model.cuda().train()
inputs = {"points": torch.from_numpy(g("xyz")).cuda(), "features": torch.from_numpy(g("feat")).cuda(), "offset": torch.from_numpy(g("offset")).cuda()}
target = torch.from_numpy(g("target")).cuda()
log = []
orig = pointops.spatial_order


def spy(idx):
    o = orig(idx)
    log.append((tuple(idx.shape), None if o is None else int(o.long().mul(torch.arange(1, o.numel() + 1, device=o.device)).sum().item() % 1000003)))
    return o


pointops.spatial_order = spy
outs = []
for r in range(4):
    log.clear()
    with torch.no_grad():
        pass
    logits, stage_list, loss, nc = M.forward_and_loss(model, crit, inputs, target)
    torch.cuda.synchronize()
    outs.append((logits.detach().clone(), loss.detach().clone(), list(log)))
for r in range(1, 4):
    print("run", r, "logits equal", torch.equal(outs[r][0], outs[0][0]), "max abs diff %.3e" % float((outs[r][0] - outs[0][0]).abs().max()),
          "loss equal", torch.equal(outs[r][1], outs[0][1]), "orders equal", outs[r][2] == outs[0][2])
print("orders of run 0:", outs[0][2][:12])
print("orders of run 1:", outs[1][2][:12])
