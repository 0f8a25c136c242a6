#!/usr/bin/env python3
"""Per-stage REAL shapes of the two reference networks (SURVEY.md §8 caveat: "benchmarks should report both the synthetic shape
and the per-stage real shapes").  Device-resident synthetic inputs, medians of HIP-event timings; one JSON document on stdout
(copied to profiles/ by the round script).

Point Transformer + CBL (pytorch side): 5 stages (n, K, C) = (40960,8,32) (10240,16,64) (2560,16,128) (640,16,256) (160,16,512),
stage point sets from this build's own FPS (stride 4), CBL nsample 36,24,24,24,24 on a 32-d latent, sub-scene labels with
kr = 4,16,64,256 from the 40960 stage-0 points, decoder interpolation k=3.
ConvNet (TF side): N = 200 000, dl0 = 0.04, density 5, 5 layers, K_lim = 26,31,38,41,39, C = 72,144,288,576,1152.
"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from contrastboundary_amd import basic_operators, heads, local_aggregation as LA, pointops, synthetic as S, tf_ops  # noqa: E402


def timeit(fn, reps=21, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def point_transformer(seed=0):
    dev = "cuda"
    xyz_np, lab_np = S.s_room(40960, seed)
    rng = np.random.default_rng(seed + 7)
    p = [torch.from_numpy(xyz_np).to(dev)]
    o = [torch.tensor([40960], dtype=torch.int32, device=dev)]
    target = torch.from_numpy(lab_np).to(dev)
    shapes = [(40960, 8, 32), (10240, 16, 64), (2560, 16, 128), (640, 16, 256), (160, 16, 512)]
    cbl_k = [36, 24, 24, 24, 24]
    rows = []
    fps_us = []
    for i in range(1, 5):                                             # TransitionDown: FPS stride 4 (blocks.py:63-67)
        no = torch.tensor([shapes[i][0]], dtype=torch.int32, device=dev)
        fps_us.append(timeit(lambda: pointops.furthestsampling(p[i - 1], o[i - 1], no), reps=5, warm=1))
        idx = pointops.furthestsampling(p[i - 1], o[i - 1], no)
        p.append(p[i - 1][idx.long()].contiguous()); o.append(no)
    stage_list = {"up": [{"p_out": p[i], "offset": o[i]} for i in range(5)]}
    for i, (n, K, C) in enumerate(shapes):
        x = torch.from_numpy(rng.normal(size=(n, C)).astype(np.float32)).to(dev)
        lat = torch.from_numpy(rng.normal(size=(n, 32)).astype(np.float32)).to(dev).requires_grad_(True)
        r = {"stage": i, "n": n, "K": K, "C": C}
        r["knnquery_us"] = timeit(lambda: pointops.knnquery_raw(K, p[i], p[i], o[i], o[i]))
        idx, _ = pointops.knnquery_raw(K, p[i], p[i], o[i], o[i])
        r["queryandgroup_us"] = timeit(lambda: pointops.queryandgroup(K, p[i], p[i], x, idx, o[i], o[i]))
        r["queryandgroup_GBps"] = (4 * n * K + 24 * n + 4 * n * C + 4 * n * K * (3 + C)) / r["queryandgroup_us"] / 1e3
        r["subtraction_us"] = timeit(lambda: pointops.subtraction(x, x, idx))                    # k_j - q_i (blocks.py:36)
        pos = torch.empty(n, K, C, device=dev).normal_(); w = torch.empty(n, K, C // 8, device=dev).normal_()
        r["aggregation_us"] = timeit(lambda: pointops.aggregation(x, pos, w, idx))               # sum_k (v+p) w (blocks.py:43)
        if i > 0:
            r["fps_from_prev_us"] = fps_us[i - 1]
            r["transition_down_knn_us"] = timeit(lambda: pointops.knnquery_raw(K, p[i - 1], p[i], o[i - 1], o[i]))
            fc = torch.empty(n, C, device=dev).normal_()
            r["interpolation_up_us"] = timeit(lambda: pointops.interpolation(p[i], p[i - 1], fc, o[i], o[i - 1]))   # blocks.py:108
            kr = 4 ** i
            r["subscene_label_kr"] = kr
            r["subscene_label_us"] = timeit(lambda: basic_operators.get_subscene_label("up", i, stage_list, target, [4, 4, 4, 4], 13))
            labels = basic_operators.get_subscene_label("up", i, stage_list, target, [4, 4, 4, 4], 13)
        else:
            labels = torch.nn.functional.one_hot(target, 13).float()
        kc = cbl_k[i]
        r["cbl_nsample"] = kc
        r["cbl_knnquery_us"] = timeit(lambda: pointops.knnquery_raw(kc, p[i], p[i], o[i], o[i], algo="set"))
        nidx, _ = pointops.knnquery_raw(kc, p[i], p[i], o[i], o[i], algo="set")

        def cbl():
            lat.grad = None
            heads.point_contrast(lat, labels, nidx, 1.0, 0.1).backward()
        r["cbl_fwd_bwd_us"] = timeit(cbl)
        rows.append({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()})
    return rows


def convnet(seed=0):
    dev = "cuda"
    xyz_np, _ = S.s_room(200000, seed, scale=4.0)
    pts = torch.from_numpy(xyz_np).to(dev); lens = torch.tensor([200000], dtype=torch.int32, device=dev)
    limits = [26, 31, 38, 41, 39]
    out = {"pyramid_us": timeit(lambda: tf_ops.segmentation_inputs_radius(pts, lens, 0.04, 5.0, 5, limits), reps=7, warm=2)}
    pyr = tf_ops.segmentation_inputs_radius(pts, lens, 0.04, 5.0, 5, limits)
    rng = np.random.default_rng(seed + 3)
    rows = []
    r0 = 0.04 * 5.0 / 2.0
    for l, C in enumerate([72, 144, 288, 576, 1152]):
        q = pyr["points"][l]; nb = pyr["neighbors"][l].contiguous()
        n, K = nb.shape
        f = torch.from_numpy(rng.normal(size=(n, C)).astype(np.float32)).to(dev)
        W = torch.empty(3, C, device=dev).normal_(); b = torch.empty(C, device=dev).normal_()
        radius = r0 * 2 ** l
        r = {"layer": l, "n": n, "K_lim": K, "C": C}
        lay_len = pyr["batches_len"][l]
        r["radius_search_us"] = timeit(lambda: tf_ops.tf_batch_neighbors(q, q, lay_len, lay_len, radius, limits[l], exact_shape=False), reps=11)
        r["adaptive_weight_us"] = timeit(lambda: LA.adaptive_weight(q, q, nb, f, radius, W, b, "mean"), reps=11)
        alg = 24 * n + 4 * n * C * 2 + 4 * n * K
        r["adaptive_weight_GBps"] = alg / r["adaptive_weight_us"] / 1e3
        r["pospool_sin_cos_us"] = timeit(lambda: LA.pospool(q, q, nb, f, radius, "sin_cos", "mean"), reps=11)
        r["pospool_xyz_us"] = timeit(lambda: LA.pospool(q, q, nb, f, radius, "xyz", "mean"), reps=11)
        if l < 4:
            pool = pyr["pools"][l].contiguous()
            r["ind_max_pool_us"] = timeit(lambda: LA.ind_max_pool(f, pool), reps=11)
        rows.append({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()})
    out["layers"] = rows
    out["pyramid_us"] = round(out["pyramid_us"], 1)
    return out


if __name__ == "__main__":
    doc = {"device": torch.cuda.get_device_name(0), "timing": "median of HIP-event timings, microseconds, device-resident inputs",
           "point_transformer_stages": point_transformer(), "convnet_N200k": convnet()}
    print(json.dumps(doc, indent=1))
