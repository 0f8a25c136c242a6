#!/usr/bin/env python3
"""bench.py — points/sec through the KNN + group + local-aggregation + CBL block on S3DIS-shaped synthetic scenes.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--points 40960] [--channels 64] [--k 16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path (contrastboundary_amd/hotpath.py) over one scene whose inputs are already
resident in HBM.  Scenes are independent, so N ranks run N scene replicas with no data-path collective (weak
scaling); the timed region is bracketed by barrier + synchronize and the MAX over ranks is used.
Rank 0 prints ONE JSON line with the driver's fields plus `roofline` (dominant kernel, HIP-event timed inside the
timed region) and `cpu_baseline` (the CPU oracle = a port of the reference's algorithm, one thread, one scene).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 matrix = f32 vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=40960)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="all stages in order on one stream")
    ap.add_argument("--no-graph", action="store_true", help="issue every step eagerly from Python instead of replaying its hipGraph")
    ap.add_argument("--no-nested", action="store_true", help="every neighbour search on its own (no derivation of K=16 from the K=36 search of the same points)")
    ap.add_argument("--side-after", default=None, help="main-stream stage after which the side stream starts (default: start of the step)")
    return ap.parse_args()


def cpu_baseline(n, c, k, seed):
    """the oracles (ports of the reference algorithms: brute-force KNN in C, the rest numpy), 1 thread, ONE full scene"""
    from tests import oracle_lib as O
    from tests import oracle_hotpath
    from contrastboundary_amd import hotpath
    sc = hotpath.Scene.synthetic_numpy(n, c, seed)
    xyz, feat, off = sc["xyz"], sc["feat"], sc["offset"]
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    parts = {}
    t0 = time.perf_counter()
    idx, _ = O.knnquery(k, xyz, xyz, off, off)
    parts["knnquery_k%d" % k] = time.perf_counter() - t0
    t1 = time.perf_counter()
    g = O.grouping_forward(np.concatenate([xyz, feat], 1), idx)
    g[..., :3] -= xyz[:, None, :]
    parts["queryandgroup"] = time.perf_counter() - t1
    parts.update(oracle_hotpath.run_rest(sc, idx, k))
    total = sum(parts.values())
    return {"value": n / total, "unit": "points/s", "cores": 1, "kind": "port",
            "sample": "1 scene of %d points (same workload), single thread; stage seconds: %s" % (
                n, {a: round(b, 3) for a, b in parts.items()})}


def main():
    args = parse()
    from contrastboundary_amd import distributed as D
    world, rank, local = D.env_world()
    torch.cuda.set_device(local)
    D.init("nccl" if world > 1 else None)                   # "nccl" is RCCL on ROCm; used for barriers / max-time only
    dist = (world > 1)

    from contrastboundary_amd import hotpath
    n, c, k = args.points, args.channels, args.k
    scene = hotpath.Scene.synthetic(n, c, seed=rank, b=1)    # every rank its own scene (weak scaling)
    stages = hotpath.stages(scene, k)
    state = {}
    # the CBL head's neighbour search (independent of the stages before it) goes to a side stream, the rest runs in order
    hints = () if args.no_nested else hotpath.search_hints(scene)
    sched = hotpath.Schedule(stages, overlap=not args.no_overlap, hints=hints)
    in_order = hotpath.Schedule(stages, overlap=False, hints=hints)

    graph = [None]

    def step(events=None):
        # steps that carry per-stage events (every EVENT_EVERY-th of the timed region) are issued eagerly, in order on one stream, so that
        # a stage's HIP-event time is that stage alone; all other steps replay the step's hipGraph (or, without it, run the schedule eagerly)
        if events is not None:
            in_order.run(state, events)
        elif graph[0] is not None:
            graph[0].replay()
        else:
            sched.run(state, None, side_after=args.side_after)

    # set-up, not part of the W warm-up steps: first touches of the library, workspaces and code objects, clock ramp of a cold device
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.5:
        step()
        torch.cuda.synchronize()
    graph_note = "eager"
    if not args.no_graph:
        # the step issues ~25 launches from Python: ~0.25-0.45 ms of host time against ~0.30 ms of device time, i.e. an eagerly issued
        # step is as fast as the host happens to be.  Captured once (same kernels, same buffers, same schedule), it is replayed with
        # ~15 us of host time per step.  Any failure to capture leaves the eager schedule in place.
        try:
            gstate = {}
            cap_stream = torch.cuda.Stream()
            cap_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap_stream):
                for _ in range(3):
                    sched.run(gstate, None, side_after=args.side_after)
            torch.cuda.current_stream().wait_stream(cap_stream)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):     # other threads (the RCCL watchdog of a multi-rank run) may call HIP
                sched.run(gstate, None, side_after=args.side_after)
            g.replay()
            torch.cuda.synchronize()
            graph[0] = g
            graph_note = "hipGraph replay of the step (torch.cuda.CUDAGraph over the C-ABI launches); steps with per-stage events issued eagerly"
        except Exception as e:                                      # noqa: BLE001 - any capture problem: stay eager
            graph[0] = None
            graph_note = "eager (graph capture failed: %s)" % type(e).__name__
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    D.barrier()
    # per-stage HIP events on every EVENT_EVERY-th step of the timed region only: 12 event records cost ~55 us, 13 % of a step
    EVENT_EVERY = 16 if args.steps >= 48 else 10 if args.steps >= 40 else 5 if args.steps >= 10 else 1      # 4 samples of the default 50 steps
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in stages] if s % EVENT_EVERY == 0 else None
          for s in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(ev[s])
    torch.cuda.synchronize()
    D.barrier()
    elapsed = D.reduce_scalar(time.perf_counter() - t0, "max")

    # per-stage average device time from the HIP events recorded inside the timed region (same stream as the launches)
    timed = [e for e in ev if e is not None]
    stage_ms = [float(np.mean([e[i][0].elapsed_time(e[i][1]) for e in timed])) for i in range(len(stages))]
    names = [st[0] for st in stages]
    # algorithmic bytes per stage; with the nested searches the first search stage does the work of both (the later request is a cache hit)
    stage_bytes = [st[2] for st in stages]
    if hints:
        first, later = names.index("knnquery_k%d" % k), names.index("cbl_knnquery_k%d" % hotpath.CBL_NSAMPLE)
        stage_bytes[first] += stage_bytes[later]
        stage_bytes[later] = 0
    gbps = lambda i: stage_bytes[i] / (stage_ms[i] * 1e-3) / 1e9
    dom = int(np.argmax(stage_ms))
    # kernel that dominates each stage (rocprofv3 --kernel-trace --stats of this same command: profiles/)
    main_kernel = {"knnquery_k16": ("knn_grid_group_kernel<16> (+ 5-launch grid build, knn_replay_kernel for the tied queries)" if args.no_nested else
                                    "the K=36 search of the same points (grid build + knn_grid_wave_kernel, which also writes the K=16 rows), knn_replay_kernel for the tied queries of both"),
                   "queryandgroup": "query_group_lds<16> (aligned 16-row pieces through LDS, cell-order schedule)", "kpconv_fwd": "kpconv_fwd_kernel (v_mfma_f32_16x16x4_f32)",
                   "cbl_knnquery_k36": "knn_grid_wave_kernel (select-then-sort, + 5-launch grid build)",
                   "cbl_mining_loss_fwd": "contrast_bwd_kernel<64,8> in fused forward+gradient mode (+ finalize)", "cbl_mining_loss_bwd": "contrast_grad_scale_kernel"}
    # `roofline`: the HBM-bound kernel of the path — the neighbour gather (north_star: ">= 50 % of the HBM roofline on the KNN-gather kernel");
    # its stage is that ONE kernel, so the stage's HIP-event time is the kernel's launch duration plus the launch gap.  The longest
    # stages (the two neighbour searches, the CBL mining kernel) are issue / latency / atomic bound, not HBM bound: they are listed under
    # `longest_stage` and in the per-stage tables with their algorithmic rates, and analysed in DESIGN.md 6.2.
    gi = names.index("queryandgroup")
    roofline = {"kernel": main_kernel["queryandgroup"], "stage": "queryandgroup", "bound": "hbm",
                "achieved": gbps(gi), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps(gi) / HBM_PEAK_GBS, "traffic": None,
                "bytes_per_launch": stages[gi][2],
                "note": "achieved = SURVEY 8(d) algorithmic bytes of the launch / its HIP-event time inside the timed region (events on the launch "
                        "stream, on the eagerly issued steps); traffic = PMC FETCH_SIZE + WRITE_SIZE per launch (profiles/r01_pmc_traffic.json)",
                "longest_stage": {"stage": names[dom], "kernel": main_kernel.get(names[dom], names[dom]), "ms": round(stage_ms[dom], 4),
                                  "algorithmic_GBps": round(gbps(dom), 1),
                                  "note": "few compulsory bytes: bound by VALU issue / memory latency / the exact replay of tied queries, not by HBM"},
                "stage_ms": {names[i]: round(stage_ms[i], 4) for i in range(len(stages))},
                "stage_algorithmic_GBps": {names[i]: (round(gbps(i), 1) if stage_bytes[i] else None) for i in range(len(stages))}}
    # HBM traffic per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes of this same command with
    # the default workload: tools/gpu_pmc.sh -> profiles/r01_pmc_traffic.json; counters cannot be read from inside the process)
    pmc = {}
    pmc_file = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc_file) and (n, c, k) == (40960, 64, 16):
        pmc = json.load(open(pmc_file))
    pmc_kernel = {"knnquery_k16": "knn_grid_group_kernel<16, true, false>", "queryandgroup": "query_group_lds<16>", "kpconv_fwd": "kpconv_fwd_kernel<true>",
                  "cbl_knnquery_k36": "knn_grid_wave_kernel<true, false>", "cbl_mining_loss_fwd": "contrast_bwd_kernel<64, 8>",
                  "cbl_mining_loss_bwd": "contrast_grad_scale_kernel"}
    traffic = lambda stage: pmc.get(pmc_kernel.get(stage, ""), {}).get("hbm_bytes_per_launch")
    roofline["traffic"] = traffic("queryandgroup")
    roofline["longest_stage"]["traffic"] = traffic(names[dom])
    if rank == 0:
        # what this device delivers on plain streams, measured here and now (after the timed region): a fill and a copy of the size of
        # the gather's output — the 8 TB/s of the spec sheet is not reachable by any kernel, these are
        probe = torch.empty(stages[gi][2] // 4, dtype=torch.float32, device="cuda"); probe2 = torch.empty_like(probe)
        def _rate(fn, nbytes, reps=10):
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record(); torch.cuda.synchronize()
            return nbytes / (a.elapsed_time(b) / reps * 1e-3) / 1e9
        fill = _rate(lambda: probe.fill_(1.0), probe.numel() * 4)
        copy = _rate(lambda: probe2.copy_(probe), 2 * probe.numel() * 4)
        roofline.update({"measured_fill_GBps": round(fill, 1), "measured_copy_GBps": round(copy, 1), "frac_of_measured_fill": gbps(gi) / fill})
        del probe, probe2
    ki = names.index("kpconv_fwd")
    roofline["mfma_kpconv"] = {"kernel": "kpconv_fwd_kernel", "bound": "mfma", "achieved": stages[ki][3] / (stage_ms[ki] * 1e-3) / 1e12,
                               "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": stages[ki][3] / (stage_ms[ki] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                               "note": "f32-input MFMA; the kernel is bound by its vector instruction stream (influence weights, operand feed), not by the matrix pipe"}

    if rank == 0:
        out = {
            "metric": "points/sec through KNN+group+KPConv+CBL block, S3DIS N=40960 K=16",
            "value": n * args.steps * world / elapsed, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "S3DIS-shaped synthetic scene (S-room), N=%d, K=%d, C=%d, 1 scene per GPU per step; stages: %s"
                       % (n, k, c, " -> ".join(s[0] for s in stages)), "parallelism": "scene-per-GPU replicas x%d (no data-path collective)" % world,
                       "issue": graph_note,
                       "schedule": (("one search per geometry: the K=%d request runs the K=%d search the CBL head declared for the same points "
                                     "(neighbor_cache hint, dropped at the end of every step) and is derived from it (cbl_knnquery_nested: the first K of each list, tied rows "
                                     "replayed); the later K=%d request is a cache hit; %s" % (k, hotpath.CBL_NSAMPLE, hotpath.CBL_NSAMPLE,
                                        "all stages in order on one stream" if args.no_overlap else
                                        "the CBL branch (needs the wide result only) on a side stream beside tie replay -> gather -> KPConv; steps with per-stage events in order"))
                                    if hints else "in order on one stream" if args.no_overlap else
                                    "two HIP streams: %s on a side stream, the other stages in order (hotpath.Schedule); the steps that carry "
                                    "per-stage events run in order" % ", ".join(hotpath.SIDE_STAGES))},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, c, k, seed=0)
        print(json.dumps(out))
    if dist:
        import torch.distributed as tdist
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
