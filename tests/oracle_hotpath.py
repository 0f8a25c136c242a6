"""CPU baseline legs of bench.py beyond KNN + group: KPConv and the CBL head through the oracles (TEST INFRASTRUCTURE)."""
import time

import numpy as np

from oracle import cbl_oracle as C
from oracle import local_aggregation_oracle as LA
from tests import oracle_lib as O


def run_rest(scene, idx, k):
    """scene = hotpath.Scene.synthetic_numpy(...) dict; returns {stage: seconds}"""
    from contrastboundary_amd import hotpath
    parts = {}
    t = time.perf_counter()
    LA.kpconv(scene["xyz"], scene["xyz"], idx, scene["feat"], scene["kernel_points"], scene["kernel_weights"], 0.12)
    parts["kpconv_fwd"] = time.perf_counter() - t
    t = time.perf_counter()
    nidx, _ = O.knnquery(hotpath.CBL_NSAMPLE, scene["xyz"], scene["xyz"], scene["offset"], scene["offset"])
    parts["cbl_knnquery_k%d" % hotpath.CBL_NSAMPLE] = time.perf_counter() - t
    t = time.perf_counter()
    C.point_contrast(scene["latent"], np.eye(13, dtype=np.float32)[scene["labels"]], nidx, temperature=1.0, weight=0.1)
    parts["cbl_mining_loss_fwd+bwd"] = time.perf_counter() - t
    return parts
