#!/usr/bin/env python3
"""Generate tests/golden/subscene_features.npz by IMPORTING the reference's own Python (Route C): RUNS ONLY IN THE BUILD CONTAINER.
get_subscene_features (pytorch/model/basic_operators.py:16-50) on arbitrary float per-point features — the general form of the sub-scene labels —
over the 5-stage pyramid of cbl_pytorch.npz's `default` case (its points and offsets are read from that fixture); `pointops.knnquery` is the CPU oracle
as in gen_cbl_goldens.py.  Stored: the features x (n0, 7) and the reference's output per stage (plus the `extend` form of stage 0 and an explicit kr)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/pytorch"


def main():
    assert os.path.isdir(REF), "needs /root/reference (build container only)"
    sys.modules["pointops_cuda"] = types.ModuleType("pointops_cuda")
    sys.path.insert(0, REF)
    from lib.pointops.functions import pointops as ref_pointops          # noqa: E402
    from model import basic_operators as ref_ops                          # noqa: E402
    from tests import oracle_lib as O

    def knnquery_cpu(nsample, xyz, new_xyz, offset, new_offset):
        idx, d2 = O.knnquery(int(nsample), xyz.numpy(), (xyz if new_xyz is None else new_xyz).numpy(), offset.numpy(), new_offset.numpy())
        return torch.from_numpy(idx), torch.sqrt(torch.from_numpy(d2))

    ref_pointops.knnquery = knnquery_cpu
    ref_ops.pointops.knnquery = knnquery_cpu
    G = np.load(os.path.join(HERE, "cbl_pytorch.npz"))
    up = [{"p_out": torch.from_numpy(G[f"default/stage{i}/p"]), "offset": torch.from_numpy(G[f"default/stage{i}/offset"])} for i in range(5)]
    sl = {"inputs": None, "up": up, "down": up}
    n0 = up[0]["p_out"].shape[0]
    x = torch.from_numpy(np.random.default_rng(11).normal(size=(n0, 7)).astype(np.float32))
    out = {"x": x.numpy()}
    nstride = torch.tensor([4, 4, 4, 4])
    for i in range(5):
        out[f"stage{i}"] = ref_ops.get_subscene_features("up", i, sl, x, nstride).numpy()
    out["stage0_extend"] = ref_ops.get_subscene_features("up", 0, sl, x, nstride, extend=True).numpy()
    out["stage2_kr5"] = ref_ops.get_subscene_features("up", 2, sl, x, nstride, kr=5).numpy()
    np.savez_compressed(os.path.join(HERE, "subscene_features.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
