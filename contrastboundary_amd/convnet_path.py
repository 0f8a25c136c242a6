"""The ConvNet side of the hot path as one measured step (BASELINE configs C3 / C5: "ScanNet-scale scene N~200k points, radius-query +
multi-scale grid subsampling stress", and the per-scene work of "S3DIS full train, ConvNet + CBL"):

    pyramid     tf_segmentation_inputs_radius, tensorflow/datasets/base.py:767-842: per layer the radius neighbours (N2,
                tf_neighbors/neighbors/neighbors.cpp:213-336, cropped to neighborhood_limits 26/31/38/41/39, config/s3dis.py:87), the grid-subsampled
                next layer (N1, grid_subsampling.cpp:114), pooling and upsampling indices: 13 radius searches + 4 grid subsamplings
    aggregation AdaptiveWeight (local_aggregation_operators.py:316-500) forward + backward on every layer at the ConvNet's widths
                C = 72, 144, 288, 576, 1152 (first_features_dim 72, config/s3dis/adapt.yaml:16-18)
    labels      scene labels of the sub-sampled layers through the pools (heads/head.py:25-49)
    CBL         contrast_head (heads/head.py:462-807; 'softnn', 'l2', sample 'label') forward + backward on the radius neighbourhoods of every layer

Shapes are data dependent (the number of voxels a layer keeps), so a step carries the host syncs the TF ops' dynamic shapes carry and is issued
eagerly; `stages()` lists the stages with the ALGORITHMIC bytes of SURVEY.md 8(d), evaluated on the sizes the step produced.
"""
import ctypes

import numpy as np
import torch

from . import heads, local_aggregation as LA, tf_ops

LIMITS = [26, 31, 38, 41, 39]          # neighborhood_limits (config/s3dis.py:87)
WIDTHS = [72, 144, 288, 576, 1152]     # local-aggregation widths per layer (first_features_dim 72, doubling)
DL0 = 0.04                             # first_subsampling_dl
DENSITY = 5.0                          # density_parameter -> r0 = dl * density / 2 = 0.1
NUM_LAYERS = 5
CBL_DIM = 32
NUM_CLASSES = 13


class ConvNetScene:
    """device-resident synthetic cloud + the trainable tensors of one AdaptiveWeight per layer + CBL latents.
    Per-layer features / latents are generated for the LARGEST possible layer size (n points) and sliced to the layer's size inside the step."""

    def __init__(self, n, seed=0, b=1, device="cuda", layers=NUM_LAYERS, widths=WIDTHS):
        a = ConvNetScene.synthetic_numpy(n, seed, b, layers, widths)
        t = lambda v: torch.from_numpy(v).to(device)
        self.n, self.seed, self.b, self.layers, self.widths = n, seed, b, layers, list(widths[:layers])
        self.points, self.lengths, self.labels = t(a["points"]), t(a["lengths"]), t(a["labels"])
        self.fc_weight = [t(w) for w in a["fc_weight"]]
        self.fc_bias = [t(w) for w in a["fc_bias"]]
        self.seeds = a["seeds"]
        self.device = device
        self._per_layer = {}

    @staticmethod
    def synthetic_numpy(n, seed=0, b=1, layers=NUM_LAYERS, widths=WIDTHS):
        from . import synthetic as S
        # the S-room scaled so that the surface density stays that of the 40960-point room
        xyz, labels = S.s_room(n, seed, scale=max(1.0, float(np.sqrt(n / 12500.0))))
        lens = np.diff(np.concatenate([[0], S.offsets(n, b, seed)])).astype(np.int32)
        rng = np.random.default_rng(seed + 500)
        fcw = [(rng.normal(size=(3, c)) * 0.5).astype(np.float32) for c in widths[:layers]]
        fcb = [rng.normal(size=(c,)).astype(np.float32) for c in widths[:layers]]
        return dict(points=xyz, lengths=lens, labels=labels.astype(np.int64), fc_weight=fcw, fc_bias=fcb, seeds=[seed + 600 + l for l in range(layers)])

    @staticmethod
    def layer_arrays_numpy(seed, rows, c):
        """features (rows,c), upstream gradient (rows,c), latent (rows,CBL_DIM) of one layer: a function of (seed, rows, c) only, so the oracle
        side of a test regenerates exactly what the device holds"""
        rng = np.random.default_rng(seed)
        return dict(feat=rng.normal(size=(rows, c)).astype(np.float32), grad=rng.normal(size=(rows, c)).astype(np.float32),
                    latent=rng.normal(size=(rows, CBL_DIM)).astype(np.float32))

    def layer_arrays(self, l, rows):
        """device copies for layer l with `rows` points (made once per size: the pyramid of a resident scene does not change)"""
        key = (l, rows)
        if key not in self._per_layer:
            a = ConvNetScene.layer_arrays_numpy(self.seeds[l], rows, self.widths[l])
            self._per_layer[key] = {k: torch.from_numpy(v).to(self.device) for k, v in a.items()}
        return self._per_layer[key]


def radius_bytes(nq, ns, limit):
    """SURVEY 8(d) N2 (+ crop): 12 Nq + 12 Ns + 4 Nq limit"""
    return 12 * nq + 12 * ns + 4 * nq * limit


def grid_bytes(n, m, b):
    """12 N read + 12 M written + 4 B"""
    return 12 * n + 12 * m + 4 * b


def adaptive_weight_bytes(n, n0, k, c):
    """SURVEY 8(d) a14 (idx given): 12 n + 12 n0 + 4 n0 C + 4 n K + 4 n C"""
    return 12 * n + 12 * n0 + 4 * n0 * c + 4 * n * k + 4 * n * c


def adaptive_weight_flops(n, k, c):
    """2 n K (3 C) + 2 n K C"""
    return 2.0 * n * k * 3 * c + 2.0 * n * k * c


def pyramid_bytes(pyr):
    """algorithmic bytes of the 13 radius searches + 4 grid subsamplings of a built pyramid"""
    total = 0
    L = len(pyr["points"])
    for l in range(L):
        n = pyr["points"][l].shape[0]
        total += radius_bytes(n, n, pyr["neighbors"][l].shape[1])
        if l + 1 < L:
            m = pyr["points"][l + 1].shape[0]
            total += grid_bytes(n, m, pyr["batches_len"][l].shape[0])
            total += radius_bytes(m, n, pyr["pools"][l].shape[1]) + radius_bytes(n, m, pyr["upsamples"][l + 1].shape[1])
    return total


class PyramidLoader:
    """The input pipeline's role (the reference builds the pyramid inside tf.data workers and prefetches, datasets/base.py:767-842 under
    tf.data.Dataset.map / prefetch): a loader thread builds the pyramid of the NEXT scene on a stream of its own while the caller issues the
    current scene's layers.  The pyramid is one native call per layer (tf_ops.segmentation_inputs_radius -> cbl_pyramid_layer), which holds no
    Python lock, so the two threads really run side by side (the op-by-op builder on a Python thread was measured SLOWER than in order:
    4.5-4.9 ms against 4.2 — the interpreter lock).  take() hands the finished pyramid to the caller's stream (event + record_stream)."""

    def __init__(self, scene, depth=1):
        """depth: pyramids in flight ahead of the consumer, each on a loader thread and stream of its own (tf.data's num_parallel_calls + prefetch).  One is
        enough here: with the layers issued by one native call the step is 3.04 ms whether one or two pyramids are in flight (two: 3.13 - 3.30 ms) — the
        device, not the loader, is what the step waits for (1.9 ms of pyramid kernels + 2.5 ms of layer kernels share it).  `waited_s` accumulates the time
        take() spent waiting for a loader thread (a consumer that is faster than the pyramid chain), so that callers can tell issue time from waiting."""
        from concurrent.futures import ThreadPoolExecutor
        self.scene = scene
        self.device = scene.points.device
        self.depth = max(1, int(depth))
        # streams with hardware queues of their own beside the consumer's (hotpath.concurrent_streams: a fresh stream can share the queue of the stream the
        # layers are issued on, and the pyramid then runs behind them instead of beside them)
        from . import hotpath
        with torch.cuda.device(self.device):
            self.streams = hotpath.concurrent_streams(self.depth, beside=[torch.cuda.current_stream(self.device)])
        self.pool = ThreadPoolExecutor(max_workers=self.depth)
        self.pending = []                                            # futures, oldest first
        self.turn = 0
        self.waited_s = 0.0

    def _build(self, stream):
        torch.cuda.set_device(self.device)
        sc, L = self.scene, self.scene.layers
        limits = LIMITS[:L]
        with torch.cuda.stream(stream):
            pyr = tf_ops.segmentation_inputs_radius(sc.points, sc.lengths, DL0, DENSITY, L, limits + [limits[-1]])
            ev = torch.cuda.Event()
            ev.record()
        return pyr, ev

    def submit(self):
        """keep `depth` pyramids in flight"""
        while len(self.pending) < self.depth:
            stream = self.streams[self.turn % self.depth]
            self.turn += 1
            stream.wait_stream(torch.cuda.current_stream(self.device))          # whatever prepared the scene
            self.pending.append(self.pool.submit(self._build, stream))

    def take(self):
        """the oldest pyramid in flight (built now if none was), usable on the caller's current stream"""
        import time
        self.submit()
        t0 = time.perf_counter()
        pyr, ev = self.pending.pop(0).result()
        self.waited_s += time.perf_counter() - t0
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for v in pyr.values():
            for t in v:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)                                        # allocated on a loader's stream, consumed on the caller's
        return pyr

    def drain(self):
        """wait until the loader threads have issued everything they were asked for (before a device-wide synchronize that closes a timed region)"""
        for f in self.pending:
            f.result()

    def close(self):
        self.drain()
        self.pending = []
        self.pool.shutdown()


class _CLayer(ctypes.Structure):
    """include/cbl_amd.h CblConvnetLayer"""
    _c = ctypes
    _fields_ = [("n", _c.c_int), ("K", _c.c_int), ("C", _c.c_int), ("Kp", _c.c_int), ("d", _c.c_int), ("radius", _c.c_float),
                ("points", _c.c_void_p), ("neighbors", _c.c_void_p), ("features", _c.c_void_p), ("fc_weight", _c.c_void_p), ("fc_bias", _c.c_void_p),
                ("grad_out", _c.c_void_p), ("latent", _c.c_void_p), ("pools", _c.c_void_p),
                ("aw_out", _c.c_void_p), ("grad_features", _c.c_void_p), ("grad_fc_weight", _c.c_void_p), ("grad_fc_bias", _c.c_void_p),
                ("cbl_loss", _c.c_void_p), ("cbl_mask", _c.c_void_p), ("grad_latent", _c.c_void_p), ("labels", _c.c_void_p)]


class NativeLayers:
    """Every layer's AdaptiveWeight forward + backward, the scene labels and the contrast head forward + backward of one scene as ONE native call
    (cbl_convnet_step, csrc/convnet_step.hip): what `stages()` issues op by op from Python — the same kernels in the same order — without the interpreter, the
    autograd engine and the allocator between the ~70 launches.  Output buffers and the workspace are kept per layer-size signature (a resident scene
    rebuilds the same pyramid; another scene's sizes get buffers of their own), so a step allocates nothing.

        run = NativeLayers(scene)
        out = run(pyr)        # -> {"aw_out": [...], "aw_grads": [(g_feat, g_w, g_b), ...], "labels": [...], "cbl_loss": [...], "cbl_mask": [...], "cbl_grad": [...]}"""

    def __init__(self, scene, temperature=1.0, weight=0.1):
        self.scene, self.temperature, self.weight = scene, float(temperature), float(weight)
        self.buffers = {}

    def _buffers(self, sizes, widths):
        key = (tuple(sizes), tuple(widths))
        hit = self.buffers.get(key)
        if hit is not None:
            return hit
        sc, dev = self.scene, self.scene.device
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        i = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
        out = {"aw_out": [], "aw_grads": [], "labels": [], "cbl_loss": [], "cbl_mask": [], "cbl_grad": []}
        for l, n in enumerate(sizes):
            c = sc.widths[l]
            out["aw_out"].append(f(n, c)); out["aw_grads"].append((f(n, c), f(3, c), f(c)))
            out["labels"].append(i(n)); out["cbl_loss"].append(f(1)); out["cbl_mask"].append(i(n)); out["cbl_grad"].append(f(n, CBL_DIM))
        hit = self.buffers[key] = {"out": out, "ws": None}
        if len(self.buffers) > 8:                                   # a handful of scenes' signatures; drop the oldest
            self.buffers.pop(next(iter(self.buffers)))
        return hit

    def __call__(self, pyr):
        from . import _lib
        sc = self.scene
        L = _lib.lib()
        nl = sc.layers
        sizes = [int(p.shape[0]) for p in pyr["points"][:nl]]
        widths = [int(nb.shape[1]) for nb in pyr["neighbors"][:nl]]
        buf = self._buffers(sizes, widths)
        out = buf["out"]
        arr = (_CLayer * nl)()
        keep = []
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        for l in range(nl):
            a = sc.layer_arrays(l, sizes[l])
            nb = pyr["neighbors"][l]
            pool = pyr["pools"][l - 1] if l > 0 else None
            keep.append((a, nb, pool))
            e = arr[l]
            e.n, e.K, e.C, e.Kp, e.d = sizes[l], widths[l], sc.widths[l], (int(pool.shape[1]) if pool is not None else 0), CBL_DIM
            e.radius = DL0 * DENSITY / 2.0 * 2 ** l
            e.points, e.neighbors, e.features = P(pyr["points"][l]), P(nb), P(a["feat"])
            e.fc_weight, e.fc_bias, e.grad_out, e.latent = P(sc.fc_weight[l]), P(sc.fc_bias[l]), P(a["grad"]), P(a["latent"])
            e.pools = P(pool) if pool is not None else None
            gf, gw, gb = out["aw_grads"][l]
            e.aw_out, e.grad_features, e.grad_fc_weight, e.grad_fc_bias = P(out["aw_out"][l]), P(gf), P(gw), P(gb)
            e.cbl_loss, e.cbl_mask, e.grad_latent, e.labels = P(out["cbl_loss"][l]), P(out["cbl_mask"][l]), P(out["cbl_grad"][l]), P(out["labels"][l])
        if buf["ws"] is None:
            L.cbl_convnet_step_workspace_bytes.restype = ctypes.c_size_t
            need = L.cbl_convnet_step_workspace_bytes(ctypes.c_int(nl), arr, ctypes.c_int(NUM_CLASSES))
            buf["ws"] = torch.empty(max(int(need), 1), dtype=torch.uint8, device=sc.device)
        ws = buf["ws"]
        labels64 = sc.labels if sc.labels.dtype == torch.int64 else sc.labels.long()
        _lib.check(L.cbl_convnet_step(ctypes.c_int(nl), arr, P(labels64), ctypes.c_int(NUM_CLASSES), ctypes.c_float(self.temperature), ctypes.c_float(self.weight),
                                      P(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(ws)), "cbl_convnet_step")
        return out


def stages(scene, backward=True, cbl=True, loader=None):
    """-> list of (name, fn(state), bytes_fn(state) -> algorithmic bytes, flops_fn(state)); fns communicate through `state`.
    loader (a PyramidLoader): the pyramid stage takes the pyramid the loader built beside the previous step and asks for the next one."""
    st = []
    L = scene.layers
    limits = LIMITS[:L]
    leaf = (lambda t: t.detach().requires_grad_(True)) if backward else (lambda t: t)

    def pyramid(s):
        if loader is not None:
            s["pyr"] = loader.take()
            loader.submit()
            return
        s["pyr"] = tf_ops.segmentation_inputs_radius(scene.points, scene.lengths, DL0, DENSITY, L, limits + [limits[-1]])
    st.append(("pyramid_radius_grid", pyramid, lambda s: pyramid_bytes(s["pyr"]), lambda s: 0.0))

    for l in range(L):
        def aw_fwd(s, l=l):
            pyr = s["pyr"]
            q, nb = pyr["points"][l], pyr["neighbors"][l]
            arr = scene.layer_arrays(l, q.shape[0])
            s["aw_in%d" % l] = (leaf(arr["feat"]), leaf(scene.fc_weight[l]), leaf(scene.fc_bias[l]))
            f, w, b = s["aw_in%d" % l]
            s["aw_out%d" % l] = LA.adaptive_weight(q, q, nb, f, DL0 * DENSITY / 2.0 * 2 ** l, w, b, "mean")

        def aw_dims(s, l=l):
            n, k = s["pyr"]["neighbors"][l].shape
            return n, k, scene.widths[l]
        st.append(("adaptive_weight_fwd_l%d" % l, aw_fwd, lambda s, d=aw_dims: adaptive_weight_bytes(d(s)[0], d(s)[0], d(s)[1], d(s)[2]),
                   lambda s, d=aw_dims: adaptive_weight_flops(*d(s))))
        if backward:
            def aw_bwd(s, l=l):
                arr = scene.layer_arrays(l, s["aw_out%d" % l].shape[0])
                s["aw_grads%d" % l] = torch.autograd.grad(s["aw_out%d" % l], s["aw_in%d" % l], arr["grad"])
            # forward's inputs + the output gradient in, feature gradient + parameter gradients out
            st.append(("adaptive_weight_bwd_l%d" % l, aw_bwd,
                       lambda s, d=aw_dims: adaptive_weight_bytes(d(s)[0], d(s)[0], d(s)[1], d(s)[2]) + 4 * d(s)[0] * d(s)[2] + 16 * d(s)[2],
                       lambda s, d=aw_dims: 2.0 * adaptive_weight_flops(*d(s))))
    if not cbl:
        return st

    def scene_labels(s):
        # labels of layer l from the labels of layer l-1 through the pooling indices ('max' = argmax of the neighbour histogram, head.py:25-49)
        lab = [scene.labels]
        for l in range(1, L):
            lab.append(heads.tf_scene_label(lab[-1], s["pyr"]["pools"][l - 1], NUM_CLASSES, "max"))
        s["labels"] = lab
    st.append(("scene_labels", scene_labels,
               lambda s: sum(4 * p.numel() + 8 * p.shape[0] + 8 * s["pyr"]["points"][l].shape[0] for l, p in enumerate(s["pyr"]["pools"][:L - 1])), lambda s: 0.0))

    for l in range(L):
        def cbl_fb(s, l=l):
            nb = s["pyr"]["neighbors"][l]
            arr = scene.layer_arrays(l, nb.shape[0])
            lat = arr["latent"].detach().requires_grad_(True)
            loss, mask = heads.tf_contrast(lat, s["labels"][l], nb, 1.0, 0.1, return_mask=True)
            s["cbl_loss%d" % l], s["cbl_mask%d" % l] = loss, mask
            s["cbl_grad%d" % l], = torch.autograd.grad(loss, lat)
        # a16 mining (idx given): 4 n K idx + 4 n d features + 4 n labels in; the gradient 4 n d out
        st.append(("tf_cbl_fwd_bwd_l%d" % l, cbl_fb, lambda s, l=l: 4 * s["pyr"]["neighbors"][l].numel() + s["pyr"]["neighbors"][l].shape[0] * (8 * CBL_DIM + 4),
                   lambda s, l=l: 1.0 * s["pyr"]["neighbors"][l].numel() * (3 * CBL_DIM + 20)))
    return st


def native_stages(scene, loader=None, runner=None):
    """the step as two stages: the pyramid (from the loader thread when there is one), then every layer's work as ONE native call (NativeLayers)"""
    L = scene.layers
    limits = LIMITS[:L]
    runner = runner if runner is not None else NativeLayers(scene)

    def pyramid(s):
        if loader is not None:
            s["pyr"] = loader.take()
            loader.submit()
            return
        s["pyr"] = tf_ops.segmentation_inputs_radius(scene.points, scene.lengths, DL0, DENSITY, L, limits + [limits[-1]])

    def layers(s):
        s["native"] = runner(s["pyr"])
    return [("pyramid_radius_grid", pyramid, lambda s: pyramid_bytes(s["pyr"]), lambda s: 0.0),
            ("layers_native_call", layers, lambda s: 0, lambda s: 0.0)]


def run_once(scene, state=None, backward=True, cbl=True, stage_list=None):
    """every stage in order on the current stream"""
    state = {} if state is None else state
    for _, fn, _, _ in (stage_list if stage_list is not None else stages(scene, backward, cbl)):
        fn(state)
    return state
