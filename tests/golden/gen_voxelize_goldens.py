#!/usr/bin/env python3
"""tests/golden/voxelize.npz from the reference's own pytorch/util/voxelize.py (imported, Route C). Build container only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/pytorch")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from util import voxelize as ref          # noqa: E402
from contrastboundary_amd import synthetic as S   # noqa: E402

out = {}
for name, (n, vs, dtype, seed) in {"f64_0p04": (6000, 0.04, np.float64, 0), "f32_0p1": (5000, 0.1, np.float32, 1)}.items():
    xyz, _ = S.s_room(n, seed=seed, voxel=0.02)
    coord = (xyz.astype(dtype) * dtype(1.0))
    coord = coord - coord.min(0)
    disc = np.floor(coord / np.array(vs, dtype=dtype))      # same expression as voxelize.py:39 with the divisor in coord's dtype
    key = ref.fnv_hash_vec(disc)
    idx_sort, count = ref.voxelize(coord, np.array(vs, dtype=dtype), mode=1)
    out[f"{name}/coord"] = coord; out[f"{name}/voxel"] = np.float64(vs); out[f"{name}/key"] = key
    out[f"{name}/idx_sort"] = idx_sort; out[f"{name}/count"] = count
np.savez_compressed(os.path.join(HERE, "voxelize.npz"), **out)
print("ok", {k: v.shape for k, v in out.items() if k.endswith("count")})
