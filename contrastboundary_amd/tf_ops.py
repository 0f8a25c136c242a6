"""Host mirror of the reference's TF-side op dispatch, /root/reference/tensorflow/ops/tf_ops.py:
    TF_OPS.get_tf_func(key) :26-73 with keys  grid_preprocess | grid | knn | radius
    tf_batch_subsampling :158-163, tf_batch_neighbors :165-168, tf_knn_search :111-129, grid_subsampling :82-103
and of the pyramid builder that calls them, /root/reference/tensorflow/datasets/base.py:767-842
(tf_segmentation_inputs_radius).  TensorFlow is not part of this path (absent here and on the GPU box): tensors are torch
CUDA tensors, stacked clouds are described by int32 per-cloud LENGTHS exactly as on the TF side.
Differences, stated once: grid-subsampled points come out in ascending-voxel-key order per cloud (the reference: libstdc++
hash order); radius neighbours of exactly equal distance are ordered by index (the reference: std::sort, unspecified)."""
import ctypes

import torch

from . import _lib, pointops

_i = ctypes.c_int
_f = ctypes.c_float
_ws = {}


def _workspace(kind, nbytes, device):
    from .neighbor_state import scratch
    return scratch(_ws, kind, nbytes, device, grow=1.25)


def _chk(t, dtype, name, dim):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous() and t.dim() == dim):
        raise TypeError(f"{name}: expected a contiguous {dim}-d CUDA tensor of {dtype}")
    return t


import threading

_offset_state = threading.local()     # per thread (the pyramid loader builds on a thread of its own): .on = nesting depth, .cache = id(lengths) -> (lengths kept alive, version, offsets)


class _offsets_cached:
    """Inside this block `_offsets` remembers the cumulative sums of the length vectors it has seen: the op-by-op pyramid builder asks for the same five
    vectors thirty times per scene.  Only there: the vectors a builder handles are written once (by the subsampling that produced them) and read afterwards —
    outside it a caller may rewrite a lengths buffer in place through the C ABI, which no tensor version counter sees, so nothing is remembered.  The state is
    thread-local: a loader thread's block neither sees nor clears what the main thread's block remembers."""

    def __enter__(self):
        _offset_state.on = getattr(_offset_state, "on", 0) + 1
        if not hasattr(_offset_state, "cache"):
            _offset_state.cache = {}

    def __exit__(self, *exc):
        _offset_state.on -= 1
        if not _offset_state.on:
            _offset_state.cache.clear()


def _offsets(lengths):
    """cumulative offsets of per-cloud lengths"""
    if not getattr(_offset_state, "on", 0):
        return torch.cumsum(lengths, 0, dtype=torch.int32)
    cache = _offset_state.cache
    hit = cache.get(id(lengths))
    if hit is not None and hit[0] is lengths and hit[1] == lengths._version:
        return hit[2]
    off = torch.cumsum(lengths, 0, dtype=torch.int32)
    cache[id(lengths)] = (lengths, lengths._version, off)
    return off


class _PendingSubsampling:
    """a grid subsampling whose kernels are enqueued and whose output size is on its way to the host (pinned buffer + event)"""
    __slots__ = ("out_p", "out_f", "out_l", "out_len", "total_host", "event", "keep")


def tf_batch_subsampling_launch(points, batches_len, sampleDl, features=None, labels=None):
    """first half of tf_batch_subsampling: enqueue the kernels and an asynchronous copy of the output size; work enqueued AFTER this call (that does not
    need the result) keeps the device busy while tf_batch_subsampling_finish waits for the size only"""
    _chk(points, torch.float32, "points", 2); _chk(batches_len, torch.int32, "batches_len", 1)
    n, b = points.shape[0], batches_len.shape[0]
    L = _lib.lib()
    dev = points.device
    fdim = features.shape[1] if features is not None else 0
    ldim = labels.shape[1] if labels is not None else 0
    if features is not None: _chk(features, torch.float32, "features", 2)
    if labels is not None: _chk(labels, torch.int32, "labels", 2)
    pend = _PendingSubsampling()
    pend.out_p = torch.empty((n, 3), dtype=torch.float32, device=dev)
    pend.out_f = torch.empty((n, fdim), dtype=torch.float32, device=dev) if fdim else None
    pend.out_l = torch.empty((n, ldim), dtype=torch.int32, device=dev) if ldim else None
    pend.out_len = torch.empty(b, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    need = L.cbl_grid_subsampling_workspace_bytes(_i(b), _i(n))
    ws = _workspace("sub", need, dev)
    offset = _offsets(batches_len)            # keep alive across the call: its raw pointer is what travels through the C ABI
    _lib.check(L.cbl_grid_subsampling(_i(b), _i(n), _lib.ptr(points), _lib.ptr(offset), _f(sampleDl), _i(fdim), _lib.ptr(features),
                                      _i(ldim), _lib.ptr(labels), _lib.ptr(pend.out_p), _lib.ptr(pend.out_f), _lib.ptr(pend.out_l), _lib.ptr(pend.out_len),
                                      _lib.ptr(total), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(points)), "cbl_grid_subsampling")
    pend.total_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
    pend.total_host.copy_(total, non_blocking=True)
    pend.event = torch.cuda.Event()
    pend.event.record()
    pend.keep = (total, offset)
    return pend


def tf_batch_subsampling_finish(pend):
    """second half: wait for the output size (data-dependent: one host wait, like the TF op's dynamic shape) and cut the outputs to it"""
    pend.event.synchronize()
    m = int(pend.total_host[0])
    res = [pend.out_p[:m], pend.out_len]
    if pend.out_f is not None: res.append(pend.out_f[:m])
    if pend.out_l is not None: res.append(pend.out_l[:m])
    return tuple(res)


def tf_batch_subsampling(points, batches_len, sampleDl, features=None, labels=None):
    """BatchGridSubsampling: (points (N,3), batches_len (B,) i32, sampleDl) -> (sub_points (M,3), sub_batches_len (B,))  tf_ops.py:158-163.
    With features (N,d) / labels (N,l) i32 also returns their per-voxel mean / majority (grid_subsampling.compute flavour)."""
    return tf_batch_subsampling_finish(tf_batch_subsampling_launch(points, batches_len, sampleDl, features, labels))


def grid_subsampling(points, features=None, labels=None, sampleDl=0.1, verbose=0):
    """cpp_subsampling.compute(points, features=, classes=, sampleDl=) on one cloud  (tf_ops.py:82-103)"""
    lens = torch.tensor([points.shape[0]], dtype=torch.int32, device=points.device)
    if labels is not None and labels.dim() == 1:
        labels = labels.view(-1, 1)
    out = tf_batch_subsampling(points, lens, sampleDl, features, labels.to(torch.int32).contiguous() if labels is not None else None)
    res = [out[0]] + list(out[2:])
    return res[0] if len(res) == 1 else tuple(res)


def tf_grid_subsampling(points, sampleDl):
    """the NON-batch TF op GridSubsampling(points, dl) -> sub_points  (tf_custom_ops/tf_subsampling/tf_subsampling.cpp:8-20 over
    grid_subsampling.cpp:6-112): the batch op on one cloud (same barycentres, same order)"""
    lens = torch.tensor([points.shape[0]], dtype=torch.int32, device=points.device)
    return tf_batch_subsampling(points, lens, sampleDl)[0]


def tf_ordered_neighbors(queries, supports, radius, limit=None):
    """the NON-batch TF op OrderedNeighbors(queries, supports, radius) -> neighbors (Nq, width) i32, ascending by distance, padded with Ns
    (tf_custom_ops/tf_neighbors/tf_neighbors.cpp:8-62 over neighbors.cpp:58-208): the batch op on one cloud"""
    q_len = torch.tensor([queries.shape[0]], dtype=torch.int32, device=queries.device)
    s_len = torch.tensor([supports.shape[0]], dtype=torch.int32, device=queries.device)
    return tf_batch_neighbors(queries, supports, q_len, s_len, radius, limit=limit)


class RadiusGrid:
    """the search grid of one support set at one radius, kept between searches: tf_batch_neighbors(..., grid=g) builds it on first use and skips the
    5-launch build afterwards.  Valid for the same supports tensor (unchanged), batches and radius, on the stream that built it."""

    def __init__(self, supports, s_batches, radius):
        self.key = (supports.data_ptr(), supports.shape[0], supports._version, s_batches.data_ptr(), float(radius))
        self.supports, self.s_batches = supports, s_batches              # keep the storage (and so the key) alive
        nbytes = _lib.lib().cbl_radius_neighbors_workspace_bytes(_i(s_batches.shape[0]), _i(supports.shape[0]))
        self.ws = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=supports.device)
        self.built = False

    def matches(self, supports, s_batches, radius):
        return self.key == (supports.data_ptr(), supports.shape[0], supports._version, s_batches.data_ptr(), float(radius))


def tf_batch_neighbors(queries, supports, q_batches, s_batches, radius, limit=None, exact_shape=True, grid=None):
    """BatchOrderedNeighbors: -> neighbors (Nq, width) i32 sorted by distance, padded with Ns  (tf_ops.py:165-168).
    limit=None: width = the largest neighbourhood, like the TF op (needs one host sync and, above 64, is unsupported);
    limit=L: the callers' crop big_neighborhood_filter (datasets/base.py:756-765) fused in; exact_shape also trims the width to
    min(L, largest neighbourhood) as the reference's slicing would ('defer': see trim_neighbor_widths); grid: a RadiusGrid of these supports."""
    _chk(queries, torch.float32, "queries", 2); _chk(supports, torch.float32, "supports", 2)
    _chk(q_batches, torch.int32, "q_batches", 1); _chk(s_batches, torch.int32, "s_batches", 1)
    nq, ns, b = queries.shape[0], supports.shape[0], q_batches.shape[0]
    L = _lib.lib()
    dev = queries.device
    lim = 64 if limit is None else int(limit)
    out = torch.empty((nq, lim), dtype=torch.int32, device=dev)
    counts = torch.empty(nq, dtype=torch.int32, device=dev)
    mc = torch.empty(1, dtype=torch.int32, device=dev)
    q_off, s_off = _offsets(q_batches), _offsets(s_batches)   # both alive until the launch is enqueued (two temporaries would share one block)
    if grid is not None:
        if not grid.matches(supports, s_batches, radius):
            raise ValueError("tf_batch_neighbors: the RadiusGrid belongs to other supports / batches / radius")
        _lib.check(L.cbl_radius_neighbors_reuse(_i(b), _i(nq), _i(ns), _lib.ptr(queries), _lib.ptr(supports), _lib.ptr(q_off), _lib.ptr(s_off),
                                                _f(radius), _i(lim), _lib.ptr(out), _lib.ptr(counts), _lib.ptr(mc), _lib.ptr(grid.ws),
                                                ctypes.c_size_t(grid.ws.numel()), _i(1 if grid.built else 0), _lib.stream_of(queries)), "cbl_radius_neighbors_reuse")
        grid.built = grid.built or nq > 0
    else:
        ws = _workspace("radius", L.cbl_radius_neighbors_workspace_bytes(_i(b), _i(ns)), dev)
        _lib.check(L.cbl_radius_neighbors(_i(b), _i(nq), _i(ns), _lib.ptr(queries), _lib.ptr(supports), _lib.ptr(q_off), _lib.ptr(s_off),
                                          _f(radius), _i(lim), _lib.ptr(out), _lib.ptr(counts), _lib.ptr(mc), _lib.ptr(ws), ctypes.c_size_t(ws.numel()),
                                          _lib.stream_of(queries)), "cbl_radius_neighbors")
    if exact_shape == "defer":
        return out, mc                         # the caller trims (trim_neighbor_widths): many searches, one host sync
    if limit is None or exact_shape:
        width = int(mc.item())
        if limit is None and width > 64:
            raise _lib.CblError(f"largest neighbourhood has {width} points; pass limit= (the reference crops to neighborhood_limits anyway)")
        out = out[:, :min(width, lim)].contiguous()
    return out


def trim_neighbor_widths(pending):
    """[(neighbors (Nq, limit), largest-neighbourhood scalar)] from tf_batch_neighbors(..., exact_shape='defer') -> the tables at the widths the
    reference's slicing gives (min(limit, largest neighbourhood)); ONE device-to-host copy for all of them"""
    if not pending:
        return []
    widths = torch.cat([mc for _, mc in pending]).tolist()
    return [out if w >= out.shape[1] else out[:, :w].contiguous() for (out, _), w in zip(pending, widths)]


def tf_knn_search(points, queries, k):
    """knn_batch(points [B,N,3], queries [B,M,3], K, omp=True) -> [B,M,K] int64 local indices  (tf_ops.py:111-129)"""
    _chk(points, torch.float32, "points", 3); _chk(queries, torch.float32, "queries", 3)
    B, N, _ = points.shape
    M = queries.shape[1]
    dev = points.device
    off = torch.arange(1, B + 1, dtype=torch.int32, device=dev) * N
    qoff = torch.arange(1, B + 1, dtype=torch.int32, device=dev) * M
    idx, _ = pointops.knnquery_raw(int(k), points.view(-1, 3), queries.view(-1, 3), off, qoff)
    out = torch.empty((B, M, int(k)), dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().cbl_knn_indices_to_local(_i(B), _i(M), _i(int(k)), _i(N), _lib.ptr(idx), _lib.ptr(out), _lib.stream_of(points)), "cbl_knn_indices_to_local")
    return out


class TF_OPS(object):
    """same lookup surface as the reference's TF_OPS.get_tf_func (tf_ops.py:26-73)"""

    @staticmethod
    def get_tf_func(key, verbose=False):
        table = {"grid_preprocess": grid_subsampling, "grid": tf_batch_subsampling, "knn": tf_knn_search, "radius": tf_batch_neighbors}
        if key not in table:
            raise NotImplementedError(f"not supported module init for key = {key}")   # 'farthest*' point at ops/sampling, absent from the reference
        return table[key]


get_tf_func = TF_OPS.get_tf_func


def _segmentation_inputs_radius_native(stacked_points, stacks_lengths, first_subsampling_dl, density_parameter, num_layers, neighborhood_limits):
    """the pyramid through cbl_pyramid: ONE native call per scene (cbl_pyramid_layer per layer inside it; the call holds no Python lock, so a loader thread
    can build the next scene's pyramid beside the thread that issues the current scene's layers — convnet_path.PyramidLoader).  Every layer's outputs are
    allocated at layer 0's capacity (a layer never has more points than the one above it) and cut to size afterwards."""
    _chk(stacked_points, torch.float32, "stacked_points", 2); _chk(stacks_lengths, torch.int32, "stacks_lengths", 1)
    L = _lib.lib()
    dev = stacked_points.device
    dl = float(first_subsampling_dl)
    r = dl * float(density_parameter) / 2.0                                  # :784-786
    n, b, nl = stacked_points.shape[0], stacks_lengths.shape[0], int(num_layers)
    lims = [int(neighborhood_limits[l]) for l in range(nl)]
    e = lambda shape, dt=torch.int32: torch.empty(shape, dtype=dt, device=dev)
    gbytes = max(int(L.cbl_radius_neighbors_workspace_bytes(_i(b), _i(n))), 1)
    grids = [e(gbytes, torch.uint8) for _ in range(nl)]
    nbs = [e((n, lims[l])) for l in range(nl)]
    pool_p = [e((n, 3), torch.float32) for _ in range(nl - 1)]
    pool_l = [e(b) for _ in range(nl - 1)]
    pools = [e((n, lims[l])) for l in range(nl - 1)]
    ups = [e((n, lims[l])) for l in range(nl - 1)]
    mcs = e(3 * nl)
    host = torch.empty(nl, dtype=torch.int32, pin_memory=True)
    ws = _workspace("pyramid", L.cbl_pyramid_layer_workspace_bytes(_i(b), _i(n)), dev)
    arr = lambda ts: (ctypes.c_void_p * max(len(ts), 1))(*[t.data_ptr() for t in ts])
    _lib.check(L.cbl_pyramid(_i(b), _i(n), _lib.ptr(stacked_points), _lib.ptr(stacks_lengths), _f(r), _f(dl), _i(nl), (ctypes.c_int * nl)(*lims),
                             arr(grids), ctypes.c_size_t(gbytes), arr(nbs), arr(pool_p), arr(pool_l), arr(pools), arr(ups), _lib.ptr(mcs),
                             ctypes.c_void_p(host.data_ptr()), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(stacked_points)), "cbl_pyramid")
    sizes = host.tolist()                                                    # the call waited for every one of them
    widths = mcs.tolist()                                                    # ONE device-to-host copy for the 13 table widths
    # Everything below layer 0 is returned as an exactly sized COPY: a view would keep its layer-0-capacity buffer alive (13 full-size tables per pyramid,
    # hundreds of MB at 100k+ points, and the loader holds two pyramids) — the deeper layers are small, so the copies are cheap; layer 0's own table is
    # already full height and is returned as it is (cut in width only where the widest neighbourhood is below the limit).
    def cut(table, rows, w):
        w = min(int(w), table.shape[1])
        if rows == table.shape[0] and w == table.shape[1]:
            return table
        return table[:rows, :w].clone(memory_format=torch.contiguous_format)
    out = {"points": [stacked_points] + [pool_p[l][:sizes[l + 1]].clone() for l in range(nl - 1)],
           "batches_len": [stacks_lengths] + pool_l,
           "neighbors": [], "pools": [], "upsamples": [torch.zeros((0, 1), dtype=torch.int32, device=dev)]}
    for l in range(nl):
        out["neighbors"].append(cut(nbs[l], sizes[l], widths[3 * l]))
        if l < nl - 1:
            out["pools"].append(cut(pools[l], sizes[l + 1], widths[3 * l + 1]))
            out["upsamples"].append(cut(ups[l], sizes[l], widths[3 * l + 2]))
    del grids, nbs, pool_p, pools, ups                                       # the capacity-sized buffers go back to the allocator with this frame
    out["pools"].append(torch.zeros((0, 1), dtype=torch.int32, device=dev))
    return out


def segmentation_inputs_radius(stacked_points, stacks_lengths, first_subsampling_dl, density_parameter, num_layers, neighborhood_limits, native=True):
    """The pyramid builder tf_segmentation_inputs_radius (datasets/base.py:767-842), geometry part: per layer the radius
    neighbours, the grid-subsampled next layer, the pooling and upsampling indices, all cropped to neighborhood_limits.
    13 radius searches + 4 grid subsamplings, on the GPU instead of single-threaded C++ inside tf.data workers.
    native (default): one cbl_pyramid call per scene; False: the same kernels issued op by op from here (identical tables)."""
    if native:
        return _segmentation_inputs_radius_native(stacked_points, stacks_lengths, first_subsampling_dl, density_parameter, num_layers, neighborhood_limits)
    with _offsets_cached():
        return _segmentation_inputs_radius_ops(stacked_points, stacks_lengths, first_subsampling_dl, density_parameter, num_layers, neighborhood_limits)


def _segmentation_inputs_radius_ops(stacked_points, stacks_lengths, first_subsampling_dl, density_parameter, num_layers, neighborhood_limits):
    """the pyramid op by op (the same kernels as the native builder, issued from here)"""
    dl = float(first_subsampling_dl)
    r = dl * float(density_parameter) / 2.0                                  # :784-786
    pts, lens = stacked_points, stacks_lengths
    out = {"points": [], "neighbors": [], "pools": [], "upsamples": [torch.zeros((0, 1), dtype=torch.int32, device=pts.device)], "batches_len": []}
    # Host synchronisation: the sub-sampled point count of every layer is data dependent (one host wait per layer, as the TF op's dynamic shape): it is
    # copied to pinned memory behind an event, the layer's self-search is enqueued behind it, and the host waits for the event only — the search is
    # still running when the next launches arrive.  The widths of the 13 neighbour tables are read back together at the end.  (A sync per search left
    # the device idle 2 of the pyramid's 3.6 ms at N = 200 000.)
    pending = []                                                             # (key, table, largest neighbourhood) in the reference's order
    # every layer's points are the supports of two or three searches at ONE radius (their own neighbourhoods, the pooling, the previous layer's
    # upsampling): 13 searches over 5 distinct grids, each built once
    grid = RadiusGrid(pts, lens, r)
    for dt in range(num_layers - 1):                                         # :795-812
        lim = int(neighborhood_limits[dt])
        sub = tf_batch_subsampling_launch(pts, lens, 2 * dl)                   # its size travels to the host while the self-search below runs
        pending.append(("neighbors", tf_batch_neighbors(pts, pts, lens, lens, r, lim, exact_shape="defer", grid=grid)))
        pool_pts, pool_lens = tf_batch_subsampling_finish(sub)
        pool_pts = pool_pts.contiguous()
        pending.append(("pools", tf_batch_neighbors(pool_pts, pts, pool_lens, lens, r, lim, exact_shape="defer", grid=grid)))
        grid = RadiusGrid(pool_pts, pool_lens, 2 * r)
        pending.append(("upsamples", tf_batch_neighbors(pts, pool_pts, lens, pool_lens, 2 * r, lim, exact_shape="defer", grid=grid)))
        out["points"].append(pts); out["batches_len"].append(lens)
        pts, lens = pool_pts, pool_lens
        r *= 2; dl *= 2
    out["points"].append(pts)                                                # :815-820
    pending.append(("neighbors", tf_batch_neighbors(pts, pts, lens, lens, r, int(neighborhood_limits[num_layers - 1]), exact_shape="defer", grid=grid)))
    for (key, _), table in zip(pending, trim_neighbor_widths([p for _, p in pending])):
        out[key].append(table)
    out["pools"].append(torch.zeros((0, 1), dtype=torch.int32, device=pts.device))
    out["batches_len"].append(lens)
    return out
