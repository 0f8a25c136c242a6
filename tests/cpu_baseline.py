"""bench.py's `cpu_baseline` leg (TEST INFRASTRUCTURE: the only part of bench.py that touches oracle/).

What is timed, on the GPU box's host cores, on ONE full scene of the bench workload:
  * "port": the CPU oracles — the reference's own algorithm (brute-force segmented KNN with its heap, knnquery_cuda_kernel.cu:65-111; gathers,
    KPConv, CBL mining as numpy) — with the KNN legs on ALL physical cores (OpenMP over the queries) and on ONE thread;
  * "reference": the reference's own CPU KNN, tensorflow/ops/nearest_neighbors/knn_.cxx:22-76 (nanoflann kd-tree: cpp_knn on one thread,
    cpp_knn_omp on all cores), compiled from /root/reference into oracle/_ref/libref_knn.so — the fastest KNN the reference itself has on a CPU.
The headline `value` is the FASTEST CPU combination (reference kd-tree KNN on all cores + the port for the stages that have no reference CPU code).
"""
import ctypes
import os
import subprocess
import time

import numpy as np


def cpu_info():
    info = {"model": "unknown", "physical_cores": os.cpu_count() or 1, "logical_cpus": os.cpu_count() or 1}
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        f = {}
        for line in txt.splitlines():
            if ":" in line:
                a, b = line.split(":", 1)
                f[a.strip()] = b.strip()
        info["model"] = f.get("Model name", "unknown")
        info["logical_cpus"] = int(f.get("CPU(s)", info["logical_cpus"]))
        info["physical_cores"] = max(1, int(f.get("Core(s) per socket", "1")) * int(f.get("Socket(s)", "1")))
    except Exception:                                                # noqa: BLE001 - lscpu absent: os.cpu_count() stands
        pass
    try:                                                             # a container may see fewer CPUs than the machine has
        info["usable_cpus"] = len(os.sched_getaffinity(0))
    except AttributeError:
        info["usable_cpus"] = info["logical_cpus"]
    return info


def _timeit(fn):
    t = time.perf_counter()
    out = fn()
    return time.perf_counter() - t, out


def run(n, c, k, seed, backward=True):
    from contrastboundary_amd import hotpath
    from oracle import cbl_oracle as C
    from oracle import local_aggregation_oracle as LA
    from tests import oracle_lib as O

    info = cpu_info()
    cores = max(1, min(info["physical_cores"], info["usable_cpus"]))
    sc = hotpath.Scene.synthetic_numpy(n, c, seed)
    xyz, feat, off = sc["xyz"], sc["feat"], sc["offset"]
    kc = hotpath.CBL_NSAMPLE
    lib = O.lib()
    P = O.P

    def port_knn(K, threads):
        idx = np.zeros((n, K), np.int32); d2 = np.zeros((n, K), np.float32)
        lib.oracle_knnquery_omp(n, K, P(xyz), P(xyz), P(off), P(off), P(idx), P(d2), threads)
        return idx

    def ref_knn(K, omp):
        R = O.ref("ref_knn")
        out = np.zeros((n, K), np.int64)
        R.ref_knn(P(xyz), ctypes.c_long(n), P(xyz), ctypes.c_long(n), ctypes.c_long(K), P(out), int(omp))
        return out

    parts = {}
    lib.oracle_knnquery_omp(min(n, 256), k, P(xyz), P(xyz), P(off), P(off), P(np.zeros((n, k), np.int32)), P(np.zeros((n, k), np.float32)), cores)   # thread pool up
    parts["knnquery_k%d_port_1thread" % k], idx = _timeit(lambda: port_knn(k, 1))
    parts["knnquery_k%d_port_allcores" % k], _ = _timeit(lambda: port_knn(k, cores))
    parts["cbl_knnquery_k%d_port_1thread" % kc], nidx = _timeit(lambda: port_knn(kc, 1))
    parts["cbl_knnquery_k%d_port_allcores" % kc], _ = _timeit(lambda: port_knn(kc, cores))
    have_ref = O.ref("ref_knn") is not None and hasattr(O.ref("ref_knn"), "ref_knn")
    if have_ref:
        os.environ["OMP_NUM_THREADS"] = str(cores)
        parts["knnquery_k%d_reference_kdtree_1thread" % k], _ = _timeit(lambda: ref_knn(k, 0))
        parts["knnquery_k%d_reference_kdtree_allcores" % k], _ = _timeit(lambda: ref_knn(k, 1))
        parts["cbl_knnquery_k%d_reference_kdtree_1thread" % kc], _ = _timeit(lambda: ref_knn(kc, 0))
        parts["cbl_knnquery_k%d_reference_kdtree_allcores" % kc], _ = _timeit(lambda: ref_knn(kc, 1))

    def group():
        g = O.grouping_forward(np.concatenate([xyz, feat], 1), idx)
        g[..., :3] -= xyz[:, None, :]
        return g
    parts["queryandgroup"], _ = _timeit(group)
    parts["kpconv_fwd"], _ = _timeit(lambda: LA.kpconv(xyz, xyz, idx, feat, sc["kernel_points"], sc["kernel_weights"], 0.12))
    parts["cbl_mining_loss_fwd+bwd"], _ = _timeit(lambda: C.point_contrast(sc["latent"], np.eye(13, dtype=np.float32)[sc["labels"]], nidx,
                                                                            temperature=1.0, weight=0.1))
    rest = ["queryandgroup", "kpconv_fwd", "cbl_mining_loss_fwd+bwd"]
    if backward:
        up = hotpath.Scene.upstream_numpy(n, c, k, seed)
        parts["queryandgroup_bwd"], _ = _timeit(lambda: O.grouping_backward(np.ascontiguousarray(up["grad_grouped"][..., 3:]), idx, n))
        parts["kpconv_bwd"], _ = _timeit(lambda: LA.kpconv_grads(xyz, xyz, idx, feat, sc["kernel_points"], sc["kernel_weights"], 0.12, up["grad_kpconv"]))
        rest += ["queryandgroup_bwd", "kpconv_bwd"]
    t_rest = sum(parts[r] for r in rest)
    t_port_1 = parts["knnquery_k%d_port_1thread" % k] + parts["cbl_knnquery_k%d_port_1thread" % kc] + t_rest
    t_port_all = parts["knnquery_k%d_port_allcores" % k] + parts["cbl_knnquery_k%d_port_allcores" % kc] + t_rest
    out = {"unit": "points/s", "cpu": info,
           "sample": "1 full scene of %d points (the bench workload, seed %d)%s; stage seconds: %s" % (
               n, seed, ", forward + backward legs" if backward else "", {a: round(b, 4) for a, b in parts.items()}),
           "port_one_thread": {"value": n / t_port_1, "cores": 1, "kind": "port"},
           "port_all_cores": {"value": n / t_port_all, "cores": cores, "kind": "port",
                              "note": "KNN legs on OpenMP threads over the queries; the numpy stages on one thread"}}
    # north_star: ">= 15x the host-CPU pointops wall-clock" is about KNN (K = 16) + group: the CPU side of that ratio, fastest variant per leg
    knn_best = min([parts["knnquery_k%d_port_allcores" % k], parts["knnquery_k%d_port_1thread" % k]] +
                   ([parts["knnquery_k%d_reference_kdtree_allcores" % k], parts["knnquery_k%d_reference_kdtree_1thread" % k]] if have_ref else []))
    out["knn_plus_group_seconds"] = {"knn_k%d_fastest_cpu" % k: knn_best, "knn_k%d_port_allcores" % k: parts["knnquery_k%d_port_allcores" % k],
                                     "queryandgroup_port": parts["queryandgroup"]}
    # headline: KNN (K = 16) + group only — the stages the reference has CPU code for (its kd-tree KNN; the gather is a copy) and the pair the
    # north star's ">= 15x host-CPU pointops" is about.  The other stages exist on a CPU only as the single-thread numpy restatements timed
    # above: a whole-block CPU number is dominated by them (KPConv alone ~80 %) and says nothing about the reference, so it is listed, not headlined.
    t_kg = knn_best + parts["queryandgroup"]
    out.update({"value": n / t_kg, "cores": cores if knn_best != parts["knnquery_k%d_port_1thread" % k] else 1, "kind": "reference" if have_ref else "port",
                "stages": "knnquery_k%d + queryandgroup" % k,
                "note": ("KNN + group only: the reference's own kd-tree KNN (knn_.cxx cpp_knn / cpp_knn_omp via oracle/_ref/libref_knn.so) or the brute-force "
                         "port, whichever is faster on this host, + the gather port" if have_ref else "oracle/_ref/libref_knn.so not present: port only")})
    t_wide = min([parts["cbl_knnquery_k%d_port_allcores" % kc]] + ([parts["cbl_knnquery_k%d_reference_kdtree_allcores" % kc],
                                                                     parts["cbl_knnquery_k%d_reference_kdtree_1thread" % kc]] if have_ref else []))
    out["whole_block"] = {"value": n / (knn_best + t_wide + t_rest), "kind": "reference KNN + port", "cores": cores,
                          "note": "every stage of the step: fastest CPU KNN for both searches + the numpy restatements (one thread) for KPConv / CBL / backward "
                                  "legs, which dominate it; not a statement about the reference"}
    return out


def run_convnet(n, seed=0, gpu_pyramid_ms=None, aw_sample=20000):
    """`cpu_baseline` of `bench.py --workload convnet`: the pyramid stage (13 radius searches + 4 grid subsamplings of one N-point cloud) through the
    reference's OWN C++ — batch_nanoflann_neighbors (neighbors.cpp:213-336) and batch_grid_subsampling (grid_subsampling.cpp:114) compiled from
    /root/reference into oracle/_ref/libref_tfops.so — on one thread, which is how the reference runs them (inside tf.data workers, one cloud batch
    per call).  Beside it, the AdaptiveWeight forward of layer 0 through the numpy restatement on a bounded sample of queries (port)."""
    from contrastboundary_amd import convnet_path as CP
    from oracle import local_aggregation_oracle as LA
    from tests import oracle_lib as O
    info = cpu_info()
    a = CP.ConvNetScene.synthetic_numpy(n, seed)
    R = O.ref("ref_tfops")
    P = O.P
    parts = {}
    if R is None:
        return {"value": None, "unit": "points/s", "cores": 1, "kind": "reference", "cpu": info, "sample": "oracle/_ref/libref_tfops.so not present: no CPU leg"}
    R.ref_batch_neighbors.restype = ctypes.c_int

    def radius(q, s, ql, sl, r, tag):
        t = time.perf_counter()
        mc = R.ref_batch_neighbors(0, len(q), P(q), len(s), P(s), len(ql), P(ql), P(sl), ctypes.c_float(r), None, ctypes.c_longlong(0))
        parts[tag] = time.perf_counter() - t
        return mc

    def grid(p, l, dl, tag):
        op = np.zeros((len(p), 3), np.float32); ol = np.zeros(len(l), np.int32)
        t = time.perf_counter()
        m = R.ref_batch_grid_subsampling(len(p), P(p), len(l), P(l), ctypes.c_float(dl), P(op), P(ol), len(p))
        parts[tag] = time.perf_counter() - t
        return op[:m].copy(), ol
    pts, lens = np.ascontiguousarray(a["points"]), np.ascontiguousarray(a["lengths"])
    r, dl = CP.DL0 * CP.DENSITY / 2.0, CP.DL0
    sizes = []
    for l in range(CP.NUM_LAYERS):
        sizes.append(len(pts))
        radius(pts, pts, lens, lens, r, "radius_l%d" % l)
        if l == CP.NUM_LAYERS - 1:
            break
        sub, sl = grid(pts, lens, 2 * dl, "grid_l%d" % l)
        radius(sub, pts, sl, lens, r, "pool_l%d" % l)
        radius(pts, sub, lens, sl, 2 * r, "upsample_l%d" % l)
        pts, lens, r, dl = sub, sl, 2 * r, 2 * dl
    t_pyr = sum(parts.values())
    # AdaptiveWeight forward, layer 0, the first `aw_sample` queries (numpy restatement, one thread)
    p0 = a["points"]
    m = min(aw_sample, n)
    idx, _, _ = O.radius_neighbors(p0[:m], p0, np.int32([m]), np.int32([n]), CP.DL0 * CP.DENSITY / 2.0, CP.LIMITS[0])
    arr = CP.ConvNetScene.layer_arrays_numpy(a["seeds"][0], n, CP.WIDTHS[0])
    t = time.perf_counter()
    LA.adaptive_weight(p0[:m], p0, idx, arr["feat"], CP.DL0 * CP.DENSITY / 2.0, a["fc_weight"][0], a["fc_bias"][0], "mean")
    t_aw = time.perf_counter() - t
    out = {"value": n / t_pyr, "unit": "points/s", "cores": 1, "kind": "reference", "cpu": info,
           "sample": "the pyramid stage of ONE cloud of %d points (seed %d; layer sizes %s) = 13 radius searches + 4 grid subsamplings through the reference's "
                     "own C++ (oracle/_ref/libref_tfops.so: nanoflann kd-tree + std::sort, hash-map grid), single thread as in the reference; stage seconds: %s"
                     % (n, seed, sizes, {k: round(v, 4) for k, v in parts.items()}),
           "pyramid_seconds": t_pyr,
           "adaptive_weight_port": {"kind": "port", "cores": 1, "queries": m, "seconds": t_aw, "queries_per_s": m / t_aw,
                                    "note": "numpy restatement of local_aggregation_operators.py:360-484 on the first %d queries of layer 0 (C = %d)" % (m, CP.WIDTHS[0])},
           "note": "value = points/s through the pyramid stage only (the stage the reference has CPU code for); the GPU step's value also carries "
                   "AdaptiveWeight and the CBL head"}
    if gpu_pyramid_ms:
        out["pyramid_speedup"] = t_pyr / (gpu_pyramid_ms * 1e-3)
    return out
