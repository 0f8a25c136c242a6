"""The measured hot path (BASELINE.json metric): one pass over one S3DIS-shaped scene of
    KNN (K=16)  ->  neighbour grouping (xyz-centred + features, (N,K,3+C))  ->  local aggregation over the K
    neighbours  ->  CBL head (neighbour search + pair mining + loss, forward and backward w.r.t. the features)
    ->  (backward=True, BASELINE config C2 "forward/backward") the backward of the block: transposed neighbour table, the
    scatter half of the grouping (K4 as a gather) and KPConv's gradients w.r.t. features and kernel weights, driven by
    fixed synthetic upstream gradients
Each stage is one or a few C-ABI launches on the current stream; `stages()` lists them with the ALGORITHMIC bytes
/ flops of SURVEY.md §8(d) so bench.py can turn a measured duration into a roofline fraction.  `Schedule` is how bench.py runs a step:
one neighbour search per geometry (the K = 16 table derived from the K = 36 search the CBL head needs on the same points, tied rows
replayed) and the CBL branch on a side stream beside the gather / KPConv branch; `run_once` is the plain in-order step with every
search on its own — the tests hold the two against each other.
"""
import os
import sys

import torch

from . import heads, local_aggregation, pointops

KP = 15                 # kernel points of the KPConv stage (KPConv's default 15; rigid kernel generator absent from the reference)
CBL_NSAMPLE = 36        # nsample[0] of the shipped CBL config (config/s3dis/origin_multi-...yaml:57)
CBL_DIM = 32            # latent width the CBL head works on (base_fdim)


class Scene:
    """device-resident synthetic scene: xyz (N,3), feat (N,C), labels (N,), offset (b,)"""

    def __init__(self, xyz, feat, labels, offset):
        self.xyz, self.feat, self.labels, self.offset = xyz, feat, labels, offset
        self.n, self.c = feat.shape

    @staticmethod
    def synthetic_numpy(n, c, seed=0, b=1):
        """host arrays: xyz, feat, labels, offset, kernel_points (KP,3), kernel_weights (KP,c), latent (n,CBL_DIM)"""
        import numpy as np
        from . import synthetic as S
        xyz, labels = S.s_room(n, seed, scale=1.0 if n <= 100000 else float(np.sqrt(n / 40000.0)))     # bigger scenes: a bigger room, same density
        rng = np.random.default_rng(seed + 1000)
        feat = rng.normal(size=(n, c)).astype(np.float32)
        off = S.offsets(n, b, seed)
        kpts = (rng.normal(size=(KP, 3)) * 0.06).astype(np.float32); kpts[0] = 0
        kw = (rng.normal(size=(KP, c)) / np.sqrt(KP)).astype(np.float32)
        latent = rng.normal(size=(n, CBL_DIM)).astype(np.float32)
        return dict(xyz=xyz, feat=feat, labels=labels, offset=off, kernel_points=kpts, kernel_weights=kw, latent=latent)

    @staticmethod
    def upstream_numpy(n, c, k, seed=0):
        """the gradients the block's backward is driven with: d loss / d grouped (n,k,3+c) and d loss / d kpconv (n,c)"""
        import numpy as np
        rng = np.random.default_rng(seed + 2000)
        return dict(grad_grouped=rng.normal(size=(n, k, 3 + c)).astype(np.float32), grad_kpconv=rng.normal(size=(n, c)).astype(np.float32))

    @staticmethod
    def synthetic(n, c, seed=0, b=1, device="cuda"):
        a = Scene.synthetic_numpy(n, c, seed, b)
        t = lambda v: torch.from_numpy(v).to(device)
        sc = Scene(t(a["xyz"]), t(a["feat"]), t(a["labels"]), t(a["offset"]))
        sc.kernel_points, sc.kernel_weights, sc.latent = t(a["kernel_points"]), t(a["kernel_weights"]), t(a["latent"])
        sc.seed = seed
        return sc

    def upstream(self, k):
        """device copies of upstream_numpy (made once per scene and k)"""
        cache = self.__dict__.setdefault("_upstream", {})
        if k not in cache:
            a = Scene.upstream_numpy(self.n, self.c, k, getattr(self, "seed", 0))
            cache[k] = {name: torch.from_numpy(v).to(self.xyz.device) for name, v in a.items()}
        return cache[k]


# stages that do not depend on the stage before them: the CBL head's neighbour search needs the coordinates only, so `run_step` issues
# it on a side stream at the start of the step, under the grid build / exact replay / gather of the main stream (those leave most
# of the device idle: a handful of small dependent launches, two workgroups replaying tied queries)
SIDE_STAGES = ("cbl_knnquery_k%d" % CBL_NSAMPLE,)


def stages(scene, k=16, backward=False, pair_tables=False):
    """-> list of (name, fn(state) -> None, algorithmic_bytes, algorithmic_flops); fns communicate through `state`.
    backward: also the backward legs of the block (BASELINE config C2).
    pair_tables (with backward): the stage that builds the CBL head's transposed K = 36 table builds the block's own K = 16 table with it (one geometry, the
    same four launches: pointops.neighbor_transpose(companion=)); the block's table stage is then a registry hit that launches nothing."""
    n, c = scene.n, scene.c
    st = []
    # leaves are made inside the step (on the stream the step runs on) and gradients are taken with torch.autograd.grad: no AccumulateGrad
    # node that outlives a step and no .grad state, so the step can be captured in a hipGraph on any stream
    leaf = (lambda t: t.detach().requires_grad_(True)) if backward else (lambda t: t)
    if backward:
        scene.upstream(k)                                             # uploaded now, not inside a step

    def knn(s):
        s["idx"], s["dist2"] = pointops.knnquery_raw(k, scene.xyz, scene.xyz, scene.offset, scene.offset)
    # compulsory 12n + 12m + 8mK bytes
    st.append(("knnquery_k%d" % k, knn, 12 * n + 12 * n + 8 * n * k, 8.0 * n * n))

    def group(s):
        s["feat_leaf"], s["kw_leaf"] = leaf(scene.feat), leaf(scene.kernel_weights)
        s["grouped"] = pointops.queryandgroup(k, scene.xyz, scene.xyz, s["feat_leaf"], s["idx"], scene.offset, scene.offset, use_xyz=True)
    # a3 fused queryandgroup: 4mK + 12n + 12m + 4nC + 4mK(3+C)
    st.append(("queryandgroup", group, 4 * n * k + 12 * n + 12 * n + 4 * n * c + 4 * n * k * (3 + c), 3.0 * n * k))

    extent = 0.12      # KP_extent 1.0 * radius 0.1*... / density (local_aggregation_operators.py:664); ~ the K=16 neighbourhood radius here

    def kpconv(s):
        s["kpconv"] = local_aggregation.kpconv(scene.xyz, scene.xyz, s["idx"], s["feat_leaf"], scene.kernel_points, s["kw_leaf"], extent)
    # a15 (idx given): 12n + 12n0 + 4n0C + 4nK + 4nC bytes; flops 2 n KP K C + 2 n KP C  (the influence weights' own
    # arithmetic, ~6 flops per (neighbour, kernel point), is not counted)
    st.append(("kpconv_fwd", kpconv, 12 * n + 12 * n + 4 * n * c + 4 * n * k + 4 * n * c, 2.0 * n * k * KP * c + 2.0 * n * KP * c))

    d = CBL_DIM

    def cbl_knn(s):
        # the CBL head of stage 0 (heads.py:185-246) searches its own neighbourhoods: nsample = 36, order-invariant consumer
        s["cbl_idx"], _ = pointops.knnquery_raw(CBL_NSAMPLE, scene.xyz, scene.xyz, scene.offset, scene.offset, algo="set")
    st.append(("cbl_knnquery_k%d" % CBL_NSAMPLE, cbl_knn, 24 * n + 8 * n * CBL_NSAMPLE, 8.0 * n * n))

    def cbl_transpose(s):
        # the CBL gradient's neighbour half is a gather over the transposed K = 36 table (no atomics): 4nK idx in, 4(n+1) + 4nK out
        s["cbl_transposed"] = pointops.neighbor_transpose(s["cbl_idx"], n, companion=s.get("idx") if (pair_tables and backward) else None)
    st.append(("cbl_neighbor_transpose", cbl_transpose, 8 * n * CBL_NSAMPLE + 4 * (n + 1), 0.0))

    def cbl_fwd(s):
        s["cbl_latent"] = scene.latent.detach().requires_grad_(True)
        # the forward needs no transposed table; the backward finds it in the registry (and waits for the stream that builds it)
        s["cbl_loss"] = heads.point_contrast(s["cbl_latent"], scene.labels, s["cbl_idx"], 1.0, 0.1)
    # a8 mining (idx given): 4nK idx + 4nd features + 4n labels in, 8n out; the latent needs a gradient, so the pass also leaves the pair
    # coefficients (4nK) and the centre half of the gradient (4nd) behind
    st.append(("cbl_mining_loss_fwd", cbl_fwd, 4 * n * CBL_NSAMPLE + 4 * n * d + 4 * n + 8 * n + 4 * n * CBL_NSAMPLE + 4 * n * d,
               1.0 * n * (CBL_NSAMPLE - 1) * (8 * d + 50)))

    def cbl_bwd(s):
        s["cbl_grad"], = torch.autograd.grad(s["cbl_loss"], s["cbl_latent"])
    # backward = the neighbour half gathered over the transposed table: 4(n+1) + 4nK table, 4nK coefficients, 4nd features,
    # 4nd centre half in, 4nd gradient out
    st.append(("cbl_mining_loss_bwd", cbl_bwd, 4 * (n + 1) + 8 * n * CBL_NSAMPLE + 12 * n * d, 3.0 * n * (CBL_NSAMPLE - 1) * d))
    if not backward:
        return st

    def transpose(s):
        s["transposed"] = pointops.neighbor_transpose(s["idx"], n)
    st.append(("neighbor_transpose_k%d" % k, transpose, 8 * n * k + 4 * (n + 1), 0.0))

    def group_bwd(s):
        # K4 (grouping_cuda_kernel.cu:16-25) for the feature columns of d loss / d grouped, as a gather over the transposed table
        s["grad_feat_group"], = torch.autograd.grad(s["grouped"], s["feat_leaf"], scene.upstream(k)["grad_grouped"])
    # 4mK (table) + 4mKC (gradient rows read) + 4nC (written)
    st.append(("queryandgroup_bwd", group_bwd, 4 * n * k + 4 * n * k * c + 4 * n * c, 1.0 * n * k * c))

    def kpconv_bwd(s):
        s["grad_feat_kpconv"], s["grad_kernel_weights"] = torch.autograd.grad(s["kpconv"], (s["feat_leaf"], s["kw_leaf"]), scene.upstream(k)["grad_kpconv"])
    # a15 backward (idx given): forward's inputs + the output gradient in, feature and kernel-weight gradients out; flops 2x the forward's
    st.append(("kpconv_bwd", kpconv_bwd, 12 * n + 12 * n + 4 * n * c + 4 * n * k + 4 * n * c + 4 * n * c + 4 * KP * c,
               2.0 * (2.0 * n * k * KP * c + 2.0 * n * KP * c)))
    return st


def pt_layer(scene, seed=0):
    """the block's PointTransformerLayer (blocks.py:14-44; C -> C, share_planes 8, nsample K) with seeded weights, on the scene's device, train mode"""
    from . import blocks
    cache = scene.__dict__.setdefault("_pt_layer", {})
    if seed not in cache:
        torch.manual_seed(1234 + seed)
        cache[seed] = blocks.PointTransformerLayer(scene.c, scene.c, 8, 16).to(scene.xyz.device).train()
    return cache[seed]


def stages_pt(scene, k=16, backward=False, pair_tables=False):
    """The block with the Point Transformer's local aggregation (BASELINE.md 3 / SURVEY 8(d): a1 + a3 + a4 + a8) instead of KPConv:
        KNN (K=16) -> relative-xyz grouping (n,K,3) (what blocks.py:36-37 gathers; the (n,K,C) gathers of x_k / x_v happen inside the fused kernels)
        -> PointTransformerLayer: q/k/v Linear, linear_p, vector attention over the K neighbours, softmax, aggregation (blocks.py:31-44)
        -> CBL head as in `stages` -> (backward) the layer's backward w.r.t. its input features and all its parameters, driven by a fixed
        upstream gradient.
    Same (name, fn, bytes, flops) tuples as `stages`; names keep the conventions Schedule / Pipeline split a step by."""
    n, c = scene.n, scene.c
    layer = pt_layer(scene)
    layer.nsample = k
    params = [p for p in layer.parameters()]
    st = []
    if backward:
        scene.upstream(k)

    def knn(s):
        s["idx"], s["dist2"] = pointops.knnquery_raw(k, scene.xyz, scene.xyz, scene.offset, scene.offset)
    st.append(("knnquery_k%d" % k, knn, 12 * n + 12 * n + 8 * n * k, 8.0 * n * n))

    def layer_fwd(s):
        s["feat_leaf"] = scene.feat.detach().requires_grad_(True) if backward else scene.feat
        s["pt_out"] = layer([scene.xyz, s["feat_leaf"], scene.offset], idx=s["idx"])
    # a4 fused PT layer (idx given), SURVEY 8(d): p 12n, x_q / x_k / x_v 12nC, idx 4nK, out 4nC;
    # flops 2 n K (9 + 3C + C^2/8 + C^2/64) + 6 n C^2
    st.append(("pt_layer_fwd", layer_fwd, 12 * n + 12 * n * c + 4 * n * k + 4 * n * c,
               2.0 * n * k * (9 + 3 * c + c * c / 8.0 + c * c / 64.0) + 6.0 * n * c * c))

    d = CBL_DIM

    def cbl_knn(s):
        s["cbl_idx"], _ = pointops.knnquery_raw(CBL_NSAMPLE, scene.xyz, scene.xyz, scene.offset, scene.offset, algo="set")
    st.append(("cbl_knnquery_k%d" % CBL_NSAMPLE, cbl_knn, 24 * n + 8 * n * CBL_NSAMPLE, 8.0 * n * n))

    def cbl_transpose(s):
        s["cbl_transposed"] = pointops.neighbor_transpose(s["cbl_idx"], n, companion=s.get("idx") if (pair_tables and backward) else None)
    st.append(("cbl_neighbor_transpose", cbl_transpose, 8 * n * CBL_NSAMPLE + 4 * (n + 1), 0.0))

    def cbl_fwd(s):
        s["cbl_latent"] = scene.latent.detach().requires_grad_(True)
        s["cbl_loss"] = heads.point_contrast(s["cbl_latent"], scene.labels, s["cbl_idx"], 1.0, 0.1)
    st.append(("cbl_mining_loss_fwd", cbl_fwd, 4 * n * CBL_NSAMPLE + 4 * n * d + 4 * n + 8 * n + 4 * n * CBL_NSAMPLE + 4 * n * d,
               1.0 * n * (CBL_NSAMPLE - 1) * (8 * d + 50)))

    def cbl_bwd(s):
        s["cbl_grad"], = torch.autograd.grad(s["cbl_loss"], s["cbl_latent"])
    st.append(("cbl_mining_loss_bwd", cbl_bwd, 4 * (n + 1) + 8 * n * CBL_NSAMPLE + 12 * n * d, 3.0 * n * (CBL_NSAMPLE - 1) * d))
    if not backward:
        return st

    def transpose(s):
        # the layer's d x_k / d x_v are gathers over the transposed K = 16 table (pt_layer.PTAttention.backward finds it in the registry)
        s["transposed"] = pointops.neighbor_transpose(s["idx"], n)
    st.append(("neighbor_transpose_k%d" % k, transpose, 8 * n * k + 4 * (n + 1), 0.0))

    def layer_bwd(s):
        grads = torch.autograd.grad(s["pt_out"], [s["feat_leaf"]] + params, scene.upstream(k)["grad_kpconv"])
        s["grad_feat_pt"], s["grad_params_pt"] = grads[0], grads[1:]
    # backward: the forward's inputs + the output gradient in, the feature gradient and the parameter gradients out; ~2x the forward's flops
    st.append(("pt_layer_bwd", layer_bwd, 12 * n + 12 * n * c + 4 * n * k + 4 * n * c + 4 * n * c + 4 * sum(p.numel() for p in params),
               2.0 * (2.0 * n * k * (9 + 3 * c + c * c / 8.0 + c * c / 64.0) + 6.0 * n * c * c)))
    return st


def search_hints(scene):
    """the widest neighbourhood each geometry of the step is searched with: the CBL head's K = 36 on the scene's own points"""
    return ((scene.xyz, CBL_NSAMPLE, "set"),)


def run_once(scene, k=16, state=None, backward=False):
    """every stage in order on the current stream (the reference's schedule)"""
    state = {} if state is None else state
    for _, fn, _, _ in stages(scene, k, backward):
        fn(state)
    return state


class Schedule:
    """The step as bench.py runs it.

        sched = Schedule(stage_list, overlap=True, hints=search_hints(scene));  sched.run(state, events=None)

    * without hints: SIDE_STAGES (the CBL head's own search) on a side stream, everything else in order on the current stream;
    * with hints (one search per geometry): the first search request runs the WIDE search and derives its own result from it; the CBL
      branch (cache hit on the wide result -> mining + loss -> backward) only needs the wide result, so with `overlap` it runs on the side
      stream next to the rest of the main branch (tie replay of the narrow search -> gather -> KPConv): an atomics-bound kernel beside an
      HBM-bound and a VALU/MFMA-bound one.

    events: optional list (one entry per stage) of (start, end) torch.cuda.Event pairs, recorded on the stream the stage runs on."""

    def __init__(self, stage_list, overlap=True, hints=(), aux_tables=True, pair_tables=False):
        """hints: (xyz, nsample, algo) triples for pointops.neighbor_cache.hint — the widest search each geometry sees during the step.
        With hints the step runs inside a neighbour cache that is dropped at the end of the step: nothing is carried from step to step.
        aux_tables: the block's own transposed table on a stream of its own right behind the search (False: in stage order on the main stream —
        measured faster for the Point Transformer block's one-step-at-a-time graph, 0.68 against 0.705 ms: a third branch in the replayed graph
        costs that block more than the 40 us of table build it takes off the backward chain)."""
        self.stage_list = stage_list
        self.pair_tables = pair_tables                              # the stage list was made with pair_tables=True (stages / stages_pt): Pipeline picks its layouts by it
        self.aux_tables = aux_tables
        self.hints = tuple(hints)
        self.overlap = overlap
        self.side = torch.cuda.Stream() if overlap else None
        self.aux = []                                               # streams of the transposed-table builds (made on first use)
        self.joined = torch.cuda.Event() if overlap else None

    def run(self, state, events=None, side_after=None):
        """side_after (schedule without hints): name of the main-stream stage after which the side stages may start (None: at the start)"""
        if self.hints:
            with pointops.neighbor_cache() as nc:
                nc.record_events = self.overlap                     # results carry the event behind their producer: the side stream waits for
                for xyz, nsample, algo in self.hints:               # the wide search alone, not for the narrow search's tie replay
                    nc.hint(xyz, nsample, algo)
                return self._run_branches(state, events) if self.overlap else self._run(state, events, None, ())
        return self._run(state, events, side_after, SIDE_STAGES if self.overlap else ())

    def _join_table(self, name, state):
        """the stream about to run a consumer of a transposed table waits for the stream that builds it — here, on the issuing thread (the
        consumer itself runs on autograd's thread and finds the wait already done: neighbor_state.transpose_lookup)"""
        key = "cbl_idx" if name.startswith("cbl_") else "idx"
        if name in ("cbl_mining_loss_bwd", "queryandgroup_bwd", "pt_layer_bwd") and key in state:
            pointops.neighbor_transpose(state[key], state[key].shape[0], build=False)

    def _launch(self, i, state, events):
        if self.overlap:
            self._join_table(self.stage_list[i][0], state)
        if events is not None:
            events[i][0].record()
        self.stage_list[i][1](state)
        if events is not None:
            events[i][1].record()

    def _run_branches(self, state, events):
        """the step as a small DAG over four streams:
            main : search -> gather -> KPConv -> [K=16 table] -> grouping backward -> KPConv backward
            side : (wide result) -> K=36 table -> CBL mining + loss -> CBL backward
            aux  : the transposed table of the block's own (K=16) neighbour table, right behind the search — a chain of small latency-bound
                   kernels that needs a few CUs, beside gather / KPConv instead of in front of its consumers
        (Every stream forks from the step's own stream: a fork of a fork — the K=36 table on a stream of its own behind the side stream —
        crashed hipGraph capture on ROCm 7.2.)
        A consumer finds its table in neighbor_state's registry and waits for the stream that built it (transpose_lookup)."""
        main = torch.cuda.current_stream()
        names = [st[0] for st in self.stage_list]
        tr_idx = [i for i, nm in enumerate(names) if "neighbor_transpose" in nm and not nm.startswith("cbl_")] if self.aux_tables else []
        side_idx = [i for i, nm in enumerate(names) if nm.startswith("cbl_")]
        main_idx = [i for i in range(len(names)) if i not in side_idx and i not in tr_idx]
        while len(self.aux) < len(tr_idx):
            self.aux.append(torch.cuda.Stream())
        self.side.wait_stream(main)                                 # the previous step is complete on every stream (joined below)
        self._launch(main_idx[0], state, events)                    # the search: wide search (event), derivation, tie replay
        before = {key: id(v) for key, v in state.items()}
        aux_of = {i: self.aux[n] for n, i in enumerate(tr_idx)}
        for i in tr_idx:                                            # the table of the block's own neighbour table: right behind the search
            aux_of[i].wait_stream(main)
            with torch.cuda.stream(aux_of[i]):
                self._launch(i, state, events)
        with torch.cuda.stream(self.side):
            for i in side_idx:                                      # cache hit on the wide result (waits for its event) -> its table -> mining + loss -> backward
                self._launch(i, state, events)
            self.joined.record(self.side)
        for i in main_idx[1:]:
            self._launch(i, state, events)
        main.wait_event(self.joined)
        for i in tr_idx:
            main.wait_stream(aux_of[i])
        for key, v in state.items():
            if before.get(key) != id(v):
                for t in (v if isinstance(v, (tuple, list)) else (v,)):
                    if torch.is_tensor(t):
                        t.record_stream(main)                       # allocated on another stream, owned by the caller from here on
        return state

    def _run(self, state, events, side_after, side_names):
        main = torch.cuda.current_stream()
        names = [st[0] for st in self.stage_list]
        side_idx = [i for i, nm in enumerate(names) if nm in side_names]
        fork_at = names.index(side_after) if (side_after is not None and side_idx) else -1
        produced = []

        def fork():
            # the side stream starts after everything already queued on the main stream: the previous step's consumers of the buffers
            # it is about to overwrite, and the inputs
            self.side.wait_stream(main)
            before = {key: id(v) for key, v in state.items()}
            with torch.cuda.stream(self.side):
                for i in side_idx:
                    self._launch(i, state, events)
                self.joined.record(self.side)
            produced.extend(v for key, v in state.items() if torch.is_tensor(v) and before.get(key) != id(v))
        if side_idx and fork_at < 0:
            fork()
        waited = not side_idx
        for i in range(len(names)):
            if i in side_idx:
                continue
            if not waited and i > side_idx[-1]:                 # first consumer of a side-stream result
                main.wait_event(self.joined)
                for v in produced:
                    v.record_stream(main)                           # allocated on the side stream, consumed here
                waited = True
            self._launch(i, state, events)
            if i == fork_at:
                fork()
        if not waited:
            main.wait_event(self.joined)
        return state


def concurrent_streams(count, candidates=12, spin_cycles=2_000_000, beside=None):
    """`count` HIP streams that really run side by side.  The runtime multiplexes streams onto a few hardware queues (4 per process by default) in
    creation order, so two fresh streams can share a queue and then execute strictly one after the other — measured: the search chain and the
    branch it was supposed to overlap landed on one queue and the pipeline ran in order.  The mapping cannot be queried, so it is observed: a
    one-thread spin kernel on two streams takes T if they have queues of their own and 2T if they share one.
    The spin kernel counts CLOCK CYCLES, and a device that was idle a moment ago (the first process on a box) is still raising its clocks: a reference time
    taken once at the start made every later pair look fast, shared queue or not (measured: the first bench run on a fresh box ran its pipeline in order,
    0.43 instead of 0.28 ms per step).  So the device is kept busy for a moment first, and every pair is compared with the SAME two kernels on ONE stream,
    issued right before it: clocks cancel out of the ratio.
    beside: stream(s) the chosen ones must ALSO run beside (e.g. the stream the caller's own work is on); not part of the result."""
    def timed(streams):
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream()
        t0.record(cur)
        for s in streams:
            s.wait_event(t0)
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(spin_cycles)
        for s in set(streams):
            cur.wait_stream(s)
        t1.record(cur)
        torch.cuda.synchronize()
        return t0.elapsed_time(t1)

    debug = os.environ.get("CBL_STREAM_PROBE_DEBUG")

    def together(a, b):
        # two trials, each bracketed by the serial reference on both sides (clocks still rising make the LATER measurement the faster one: the pair
        # is compared with the faster of the two references); a pair counts as concurrent only if it is in both trials
        for _ in range(2):
            s0 = timed([a, a]); pair = timed([a, b]); s1 = timed([a, a])
            if debug:
                print("probe: serial %.3f pair %.3f serial %.3f ms -> %s" % (s0, pair, s1, pair < 0.75 * min(s0, s1)), file=sys.stderr)
            if not pair < 0.75 * min(s0, s1):
                return False
        return True
    pool = [torch.cuda.Stream() for _ in range(candidates)]
    if not hasattr(torch.cuda, "_sleep"):                           # no spin kernel to observe with: take the streams as they come
        return pool[:count]
    # keep the device busy until two consecutive measurements of the same kernel agree (the clocks are up), at most ~0.5 s
    last = None
    for _ in range(60):
        for _ in range(8):
            torch.cuda._sleep(spin_cycles)
        t = timed([pool[0]])
        if last is not None and abs(t - last) < 0.02 * last:
            break
        last = t
    fixed = [] if beside is None else (list(beside) if isinstance(beside, (list, tuple)) else [beside])
    chosen = []
    for s in pool:
        if len(chosen) == count:
            break
        if all(together(c, s) for c in fixed + chosen):
            chosen.append(s)
    if len(chosen) < count:                                         # fewer independent queues than asked for: the rest share
        chosen += [p for p in pool if p not in chosen][:count - len(chosen)]
    return chosen


class Pipeline:
    """Consecutive steps software-pipelined over four HIP streams, every segment of a step replayed from a linear hipGraph of its own.  Default layout
    ("split_t36_first", round 4):

        search : nothing but the searches, one step after the other (grid build, wide search, tie replay: a chain of mostly small
                 latency-bound launches, a third of an in-order step, during which most of the GPU idles)          -> event `found`
        fwd    : gather -> KPConv (or the attention layer's forward)                                    behind `found`, -> event `fdone`
        bwd    : K=16 table -> grouping backward -> KPConv backward (or the layer's backward)                      behind `fdone`
        side   : K=36 table -> CBL mining + loss -> CBL backward                                                   behind `found`

    so the search of step i+1 runs beside the forward kernels of step i and the backward kernels of step i-1: every table is built on the
    stream that consumes it (no table stream, no table events), and no stream carries more than ~150 us of kernels per step.  The ORDER inside the CBL
    chain matters: with the pair kernel first ("split") the pipeline has two stable alignments, 0.270 and 0.290 ms per step, picked by the timing of the
    first steps after an idle device; with the table's small kernels first every run is the fast one.  "alt_bwd" (the Point Transformer
    block, whose backward chain is 0.46 of its 0.55 ms): consecutive steps' backward chains on two streams in turn, searches and CBL chain sharing one.
    "tables" is round 2's layout, kept as the baseline the two defaults are measured against: tables on a stream of their own, forward and backward of the
    block on one stream (`rest`), which serialises the backward of step i with the forward of step i+1: 0.310 ms per step against 0.272 for the KPConv block,
    0.68 (one step at a time was faster) against 0.505 ms for the Point Transformer block, same box, same kernels.
    Steps are independent scenes (in bench.py: the same resident scene); a step writes into one of SLOTS slots (its neighbour tables, orders,
    outputs: self.states[slot]) and the search of step i+SLOTS waits for every other stream's part of step i before it overwrites their slot.  A
    step runs inside its own neighbour cache, which only exists while the step is captured.

    Why linear graphs on real streams and not one graph with branches: a replayed hipGraph is spread over at most four hardware queues by the
    runtime (ROCm 7.2; DEBUG_HIP_FORCE_GRAPH_QUEUES above 4 aborts), and which branch lands on which queue is the runtime's choice — measured on a
    10-step graph, every second step's search was put on the queue of the previous step's backward kernels, which serialised exactly what this
    schedule overlaps (step period alternating 190 / 490 us).  A linear graph stays on the queue of the stream it is launched on, and the streams
    are chosen so that they have queues of their own (concurrent_streams; a fifth stream shares a queue: GPU_MAX_HW_QUEUES=8 made the step
    slower, 0.37-0.47 ms).  Graphs do not share a memory pool: they run concurrently."""

    SLOTS = 3
    LAYOUT = "split_t36_first"                                            # which of the layouts below; CBL_PIPELINE_LAYOUT / CBL_PIPELINE_SLOTS override (experiments)

    @staticmethod
    def plan(names, layout):
        """-> (stream names, segments) of `layout` for a step whose stages are called `names` (hotpath.stages / stages_pt conventions); a segment is
        (name, stream, stage indices in issue order, events waited for, event recorded behind it), segments in issue order.  Pure: no device needed
        (tests/test_host_pipeline_layouts.py checks every layout's coverage and ordering)."""
        ix = lambda pred: [i for i, nm in enumerate(names) if pred(nm)]
        search = ix(lambda nm: "knnquery" in nm)                    # the block's search, then the CBL head's request (cache hit)
        t16 = ix(lambda nm: "neighbor_transpose" in nm and not nm.startswith("cbl_"))
        t36 = ix(lambda nm: "neighbor_transpose" in nm and nm.startswith("cbl_"))
        cbl = [i for i in ix(lambda nm: nm.startswith("cbl_")) if i not in search and i not in t36]
        block = [i for i in range(len(names)) if i not in search + t16 + t36 + cbl]
        bwd = lambda i: names[i].endswith("_bwd")
        fwd_b, bwd_b = [i for i in block if not bwd(i)], [i for i in block if bwd(i)]
        fwd_c, bwd_c = [i for i in cbl if not bwd(i)], [i for i in cbl if bwd(i)]
        # (name, stream, stages in issue order, events waited for, event recorded behind it) in issue order
        if layout == "tables":
            streams = ("search", "tables", "rest", "side")
            segs = [("search", "search", search, (), "found"),
                    ("t16", "tables", t16, ("found",), "t16"),
                    ("fwd", "rest", fwd_b, ("found",), None),
                    ("cblfwd", "side", fwd_c, ("found",), None),
                    ("t36", "tables", t36, ("found",), "t36"),
                    ("bwd", "rest", bwd_b, ("t16",) if t16 else (), None),
                    ("cblbwd", "side", bwd_c, ("t36",) if t36 else (), None)]
        elif layout == "split_t36_first":
            # The default: the block's backward on a stream of its own behind its table, so that the backward kernels of step i run beside the forward kernels of
            # step i+1; every table is built on the stream that consumes it (no table stream, no table events); the K=36 table IN FRONT of the CBL forward: with
            # the pair kernel first the pipeline had two stable regimes, 0.270 and 0.290 ms per step, picked by the timing of the first steps after an idle
            # device; with the table's small kernels — not the full-device pair kernel — beside the start of the next wave search and the gather, every run
            # is the fast one (round 4: 12 of 12; the six other orders and a timed re-dealing of chains to streams that were measured then are in the history
            # of this file and in DESIGN.md 6.4).
            streams = ("search", "fwd", "bwd", "side")
            segs = [("search", "search", search, (), "found"),
                    ("fwd", "fwd", fwd_b, ("found",), "fdone"),
                    ("cbl", "side", t36 + fwd_c + bwd_c, ("found",), None),
                    ("bwd", "bwd", t16 + bwd_b, ("fdone",), None)]
        elif layout == "alt_bwd":
            # for a block whose backward chain is longer than everything else of the step together (the Point Transformer block: 0.46 of 0.56 ms): consecutive
            # steps' backward chains on TWO streams in turn ("bwd*": by step parity), so that they overlap; the searches and the CBL chain share the fourth
            # stream.  PT block 0.549 -> 0.505 ms (four slots; the CBL chain on the forward stream instead: 0.55); the KPConv block LOSES with it (0.348)
            streams = ("search", "fwd", "bwd0", "bwd1")
            segs = [("search", "search", search, (), "found"),
                    ("fwd", "fwd", fwd_b + t16, ("found",), "fdone"),
                    ("cbl", "search", t36 + fwd_c + bwd_c, (), None),
                    ("bwd", "bwd*", bwd_b, ("fdone",), None)]
        elif layout == "pair_split":
            # "split_t36_first" for stage lists whose K = 36 table stage builds the block's K = 16 table with it (stages(pair_tables=True)): the pair build is a
            # graph of its own at the head of the CBL chain, the event behind it is what the backward chain waits for besides the forward's
            streams = ("search", "fwd", "bwd", "side")
            segs = [("search", "search", search, (), "found"),
                    ("fwd", "fwd", fwd_b, ("found",), "fdone"),
                    ("tabs", "side", t36, ("found",), "tabs"),
                    ("cbl", "side", fwd_c + bwd_c, (), None),
                    ("bwd", "bwd", t16 + bwd_b, ("fdone", "tabs") if t36 else ("fdone",), None)]
        elif layout == "pair_alt_bwd":
            # "alt_bwd" with the pair build at the head of the CBL chain (on the search stream)
            streams = ("search", "fwd", "bwd0", "bwd1")
            segs = [("search", "search", search, (), "found"),
                    ("fwd", "fwd", fwd_b, ("found",), "fdone"),
                    ("tabs", "search", t36, (), "tabs"),
                    ("cbl", "search", fwd_c + bwd_c, (), None),
                    ("bwd", "bwd*", t16 + bwd_b, ("fdone", "tabs") if t36 else ("fdone",), None)]
        else:
            raise ValueError("unknown pipeline layout %r" % layout)
        return streams, [sg for sg in segs if sg[2]]

    def __init__(self, sched, layout=None, slots=None):
        import os
        assert sched.hints, "the pipeline is built on the one-search-per-geometry schedule"
        self.sched = sched
        self.layout = layout or os.environ.get("CBL_PIPELINE_LAYOUT", self.LAYOUT)
        self.SLOTS = int(slots or os.environ.get("CBL_PIPELINE_SLOTS", self.SLOTS))
        if self.layout.startswith("pair_") != bool(getattr(sched, "pair_tables", False)):
            # a plain layout runs the block's table stage on a stream that is not ordered behind the stage that built the pair (and a pair layout's extra
            # event orders nothing the plain stages need): refuse instead of racing
            raise ValueError("pipeline layout %r does not fit a step whose stages were made with pair_tables=%s" % (self.layout, bool(getattr(sched, "pair_tables", False))))
        self.STREAMS, self.segments = self.plan([st[0] for st in sched.stage_list], self.layout)
        self.streams = dict(zip(self.STREAMS, concurrent_streams(len(self.STREAMS))))        # streams with hardware queues of their own
        self.states = [{} for _ in range(self.SLOTS)]
        self.graphs = [dict() for _ in range(self.SLOTS)]
        recorded = sorted({sg[4] for sg in self.segments if sg[4]})
        self.events = [{nm: torch.cuda.Event() for nm in recorded} for _ in range(self.SLOTS)]
        self.done = [{c: torch.cuda.Event() for c in self.STREAMS[1:]} for _ in range(self.SLOTS)]
        self.count = 0

    def describe(self):
        """the layout in words (bench.py's `config.issue`)"""
        chains = " | ".join("%s: %s" % (sg[1], "+".join(self.sched.stage_list[i][0] for i in sg[2])) for sg in self.segments)
        return ("layout '%s': one linear graph per segment on %d streams (stream: stages — %s), consecutive steps software-pipelined over %d output slots"
                % (self.layout, len(self.STREAMS), chains, self.SLOTS))

    def _segment(self, seg, state):
        for i in seg[2]:
            self.sched.stage_list[i][1](state)
        if seg[0] == "search":
            state["_order"] = pointops.spatial_order(state["idx"])                          # kept alive for the branches

    def capture(self):
        from . import neighbor_state
        for slot in range(self.SLOTS):
            for graphed in (False, True):                           # once eagerly (workspaces of the streams, code objects), then captured
                state = {}
                with pointops.neighbor_cache() as nc, neighbor_state.streams_ordered_by_caller(self.streams.values()):
                    nc.record_events = False
                    for xyz, nsample, algo in self.sched.hints:
                        nc.hint(xyz, nsample, algo)
                    for seg in self.segments:                       # in dependency order, the device idle between two segments
                        stream = self.streams[seg[1].replace("*", "0")]
                        torch.cuda.synchronize()
                        if graphed:
                            g = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                                self._segment(seg, state)
                            self.graphs[slot][seg[0]] = g
                        else:
                            with torch.cuda.stream(stream):
                                self._segment(seg, state)
                torch.cuda.synchronize()
            self.states[slot] = state
        self.count = 0
        for _ in range(2 * self.SLOTS):                             # every graph has run, in pipeline order
            self.step()
        torch.cuda.synchronize()

    def step(self):
        """issue one step: one graph launch per segment; returns at once"""
        slot = self.count % self.SLOTS
        S, ev = self.streams, self.events[slot]
        if self.count >= self.SLOTS:
            for c in self.STREAMS[1:]:
                S["search"].wait_event(self.done[slot][c])          # the step that used this slot last is through with it
        else:
            S["search"].wait_stream(torch.cuda.current_stream())    # whatever prepared the inputs
        self.count += 1
        for name, sname, _, waits, record in self.segments:
            stream = S[sname.replace("*", str((self.count - 1) & 1))]  # "bwd*": two streams in turn
            for w in waits:
                stream.wait_event(ev[w])
            with torch.cuda.stream(stream):
                self.graphs[slot][name].replay()
                if record is not None:
                    ev[record].record()
        for c in self.STREAMS[1:]:
            self.done[slot][c].record(S[c])

    def join(self):
        """the caller's stream waits for everything issued so far"""
        cur = torch.cuda.current_stream()
        for c in self.STREAMS:
            cur.wait_stream(self.streams[c])
