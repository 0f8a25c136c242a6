"""radius search (cbl_radius_neighbors, row a11) alone at the ConvNet workload's layer-0 / layer-1 shapes: python tools/radius_time.py -> one JSON line"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrastboundary_amd import synthetic as S, tf_ops  # noqa: E402


def timed(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    xyz, _ = S.s_room(200000, seed=0, scale=float(np.sqrt(200000 / 40000.0)))
    x = torch.from_numpy(xyz).cuda(); lens = torch.tensor([200000], dtype=torch.int32, device="cuda")
    out = {}
    g0 = tf_ops.RadiusGrid(x, lens, 0.1)
    out["l0_self_r0.1_lim26_us"] = round(timed(lambda: tf_ops.tf_batch_neighbors(x, x, lens, lens, 0.1, 26, exact_shape=False, grid=g0)), 1)
    sub, sl = tf_ops.tf_batch_subsampling(x, lens, 0.08)
    sub = sub.contiguous()
    out["l0_pool_us"] = round(timed(lambda: tf_ops.tf_batch_neighbors(sub, x, sl, lens, 0.1, 26, exact_shape=False, grid=g0)), 1)
    g1 = tf_ops.RadiusGrid(sub, sl, 0.2)
    out["l1_self_r0.2_lim31_us"] = round(timed(lambda: tf_ops.tf_batch_neighbors(sub, sub, sl, sl, 0.2, 31, exact_shape=False, grid=g1)), 1)
    out["l1_up_us"] = round(timed(lambda: tf_ops.tf_batch_neighbors(x, sub, lens, sl, 0.2, 26, exact_shape=False, grid=g1)), 1)
    out["n1"] = int(sub.shape[0])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
