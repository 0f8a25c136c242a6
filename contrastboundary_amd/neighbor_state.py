"""State that the op wrappers of pointops.py share across calls: the per-(device, stream) workspace, the processing-order registry
(cell order of a self-search, kept per geometry and per stream) and the per-forward neighbour cache (SURVEY.md 8(f) rank 1:
/root/reference/pytorch/model/pointtransformer_seg.py asks for the same knnquery 57 times per forward).  No op bodies here: pointops.py
holds the wrappers, this module holds what they remember."""
import collections
import threading

import torch

_ws_cache = {}


def scratch(cache, key, nbytes, device, grow=1.0):
    """Scratch bytes for one launch on the current stream.  Eagerly: one buffer per (key, device, stream handle), grown on demand and reused — launches
    on a stream are ordered, so they may share it.  While the stream is CAPTURING: a fresh buffer per call.  A captured launch bakes the address
    in, and the graph is later replayed on whatever stream its owner chooses, concurrently with eager work on other streams — torch hands out its
    32 pooled streams round-robin, so a stream object created later can carry the very handle the graph was captured on and would share (race
    on) or regrow (free) the cached buffer under the replaying graph: seen as an illegal address in the graphed training step once enough tests
    had drawn streams from the pool.  Allocated during capture the buffer belongs to the graph's private pool for the graph's lifetime."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
    k = (key, device, torch.cuda.current_stream(device).cuda_stream)
    ws = cache.get(k)
    if ws is None or ws.numel() < nbytes:
        ws = cache[k] = torch.empty(int(nbytes * grow) + 256, dtype=torch.uint8, device=device)
    return ws


def _workspace(nbytes, device):
    """per-(device, stream) scratch for the grid KNN; grown on demand, reused across calls (see `scratch`)"""
    if nbytes == 0:
        return None
    return scratch(_ws_cache, "knn", nbytes, device, grow=1.25)


# ------------------------------------------------------------------------------------------------ processing order
# The grid build of a self-search sorts the supports into cells; that sequence ("cell order") is a spatially coherent processing
# order for every kernel that walks the points and gathers their neighbours.  It is kept per geometry (the coordinate tensor) and per
# stream, and handed to the *_ordered C entry points: the VALUES never depend on it — a stale or missing order only costs locality.
ORDER_MIN_POINTS = 8192         # below this the tables sit in L2 anyway
_order_registry = collections.OrderedDict()     # (data_ptr, n, version, device) -> {stream id: (order tensor, stream that holds it)}
_ORDER_REGISTRY_MAX = 16


def _version_of(t):
    try:
        return t._version
    except RuntimeError:                                            # inference tensors do not track a version counter
        return -1


def _order_key(points):
    return (points.data_ptr(), points.shape[0], _version_of(points), points.device)


def _order_wanted(points, stream_id):
    """does a self-search over `points` on this stream still have to produce the cell order?"""
    if points.shape[0] < ORDER_MIN_POINTS or not use_spatial_order:
        return False
    ent = _order_registry.get(_order_key(points))
    return ent is None or stream_id not in ent


def _order_register(points, order, stream):
    key = _order_key(points)
    ent = _order_registry.setdefault(key, {})
    ent[stream.cuda_stream] = (order, stream)                      # no event here (a record costs ~4 us of stream time): see spatial_order
    _order_registry.move_to_end(key)
    while len(_order_registry) > _ORDER_REGISTRY_MAX:
        _order_registry.popitem(last=False)
    cache = neighbor_cache.active()
    if cache is not None:                                            # a cached pass owns what it registers: dropped with the cache
        cache.order_keys.append(key)
        # ... and keeps the tensor alive as long as it lives itself: the registry is a bounded LRU, and a consumer may have baked the
        # order's address into a captured hipGraph (geometry.StaticGeometry keeps its cache for exactly that long)
        cache.order_refs.append(order)


def _order_alias(idx, points):
    """the neighbour table of a self-search over `points` shares their processing order (ops that are given idx but no coordinates)"""
    ent = _order_registry.get(_order_key(points))
    if ent:
        _order_registry[_order_key(idx)] = ent
        while len(_order_registry) > _ORDER_REGISTRY_MAX:
            _order_registry.popitem(last=False)
        cache = neighbor_cache.active()
        if cache is not None:
            cache.order_keys.append(_order_key(idx))


class streams_ordered_by_caller:
    """inside: lookups of processing orders and transposed tables made ON THE GIVEN STREAMS hand out what they have without ordering the current stream
    behind the stream that produced it — for callers that sequence those streams themselves (hotpath.Pipeline captures every chain of a step as a
    hipGraph of its own; a wait on another stream's live event has no place inside such a capture).  Scoped by stream handle, not by thread: autograd's
    thread runs a backward op on the stream of its forward op and must see the same answer, while other threads' work on other streams (a prefetch, a
    second trainer) keeps its wait_stream / record_stream."""
    _streams = collections.Counter()                                # cuda_stream handle -> nesting depth

    def __init__(self, streams):
        self.handles = [s.cuda_stream for s in streams]

    def __enter__(self):
        for h in self.handles:
            streams_ordered_by_caller._streams[h] += 1
        return self

    def __exit__(self, *exc):
        for h in self.handles:
            streams_ordered_by_caller._streams[h] -= 1
            if streams_ordered_by_caller._streams[h] <= 0:
                del streams_ordered_by_caller._streams[h]
        return False

    @staticmethod
    def applies(stream):
        return stream.cuda_stream in streams_ordered_by_caller._streams


def spatial_order(points):
    """-> int32 (n,) processing order of `points` (cell order of an earlier self-search over the same tensor; also keyed by the
    neighbour table that search returned), or None"""
    if not use_spatial_order or points.shape[0] < ORDER_MIN_POINTS:
        return None
    ent = _order_registry.get(_order_key(points))
    if not ent:
        return None
    cur = torch.cuda.current_stream(points.device)
    hit = ent.get(cur.cuda_stream)
    if hit is not None:
        return hit[0]
    order, producer = next(iter(ent.values()))                   # produced on another stream: order after it, keep it alive for this one
    if streams_ordered_by_caller.applies(cur):
        return order
    cur.wait_stream(producer)
    order.record_stream(cur)
    ent[cur.cuda_stream] = (order, cur)                             # ordered behind the producer from here on
    return order


use_spatial_order = True


class natural_order:
    """inside: no processing orders — every point-walking kernel takes the points in index order, and no search exports its cell order.  Values never depend
    on an order EXCEPT through summation order: csrc/pt_layer.hip sums its BatchNorm statistics tile by tile in processing order, and which search of a geometry
    ran first (neighbour cache or not, geometry prefetch or not) decides that order.  Two runs that must be compared bit for bit (tests/test_gpu_model.py) run
    in here; costs the gathers their L2 locality, nothing else."""

    def __enter__(self):
        global use_spatial_order
        self.was = use_spatial_order
        use_spatial_order = False
        return self

    def __exit__(self, *exc):
        global use_spatial_order
        use_spatial_order = self.was
        return False


# ------------------------------------------------------------------------------------------------ transposed neighbour tables
# cbl_neighbor_transpose of a neighbour table, kept per table.  A module-level registry (not the thread-local cache): backward passes run on
# autograd's device thread and must find what the forward thread built.  Unlike a processing order the VALUES of a consumer depend on
# the table, so an entry keeps the neighbour table itself alive (its storage cannot be recycled under the key) and is only served to the
# same storage at the same version.
_transpose_registry = collections.OrderedDict()     # _order_key(idx) -> (idx, order or None, inv_start, inv_src, producer stream, streams that waited)
_TRANSPOSE_REGISTRY_MAX = 64                    # one training step of the reference network uses ~25 distinct tables (5 stages x {blocks, CBL, interpolation, ...})


def transpose_lookup(idx):
    """-> (order or None, inv_start, inv_src) registered for this neighbour table, or None"""
    if _version_of(idx) < 0:
        return None
    ent = _transpose_registry.get(_order_key(idx))
    if ent is None or ent[0].data_ptr() != idx.data_ptr() or ent[0].shape != idx.shape:
        return None
    _, order, inv_start, inv_src, producer, waited = ent[:6]
    cur = torch.cuda.current_stream(idx.device)
    if producer != cur and cur.cuda_stream not in waited and not streams_ordered_by_caller.applies(cur):   # built on another stream: order after it ONCE, keep the tensors alive for this one
        cur.wait_stream(producer)
        for t in (inv_start, inv_src) + (() if order is None else (order,)):
            t.record_stream(cur)
        waited.add(cur.cuda_stream)
    return order, inv_start, inv_src


def transpose_register(idx, order, inv_start, inv_src):
    if _version_of(idx) < 0:
        return
    key = _order_key(idx)
    cache = neighbor_cache.active()
    _transpose_registry[key] = (idx, order, inv_start, inv_src, torch.cuda.current_stream(idx.device), set(), cache is not None)
    _transpose_registry.move_to_end(key)
    while len(_transpose_registry) > _TRANSPOSE_REGISTRY_MAX:
        _transpose_registry.popitem(last=False)
    if cache is not None:                                            # a cached pass owns what it registers: dropped with the cache
        cache.transpose_keys.append(key)


def release_unowned_transposes():
    """Tables built outside any neighbour cache — in a backward pass: autograd's thread has no active cache, and the forward's `with` has
    exited by then — are only bounded by the registry's LRU; each pins its neighbour table, inv_start, inv_src and order (~12 bytes per pair).
    A training loop calls this once per step, behind backward (train_step.DataParallelTrainer.step does); returns the number dropped."""
    dead = [k for k, ent in _transpose_registry.items() if not ent[6]]
    for k in dead:
        _transpose_registry.pop(k, None)
    return len(dead)



class neighbor_cache:
    """Per-forward neighbour-index cache.  The reference's network asks for the SAME neighbour search
    many times per forward — every PointTransformerLayer of a stage runs knnquery(nsample, p, p, o, o) twice (blocks.py:34-35), the
    decoder blocks again, 57 launches in total — because each pointops call is self-contained.  Inside

        with pointops.neighbor_cache() as nc:
            logits, stage_list = model(inputs); loss = criterion(logits, target, stage_list)

    identical requests (same nsample, same coordinate / offset tensors by storage, shape and version) are answered from the
    first result; nothing else changes, so modules keep the reference's signatures.  An 'auto' (reference-order) result also
    serves a later 'set' request.  The cache holds references to the keyed tensors, so storage cannot be recycled under it; it
    is dropped when the context exits.  `nc.hits` / `nc.misses` count requests."""
    _tls = threading.local()             # the active cache is per thread (nn.DataParallel replicas run the mirrors from worker threads)

    def __init__(self):
        self.store, self.hits, self.misses = {}, 0, 0
        self.host = {}                                              # host copies of offset tensors, see host_offsets()
        self.hints = {}                                             # geometry -> (widest nsample it will be searched with, algo), see hint()
        self.wide = {}                                              # geometry -> (nsample, algo) of the widest result in the store
        self.derived = 0                                            # requests answered from a wider result (cbl_knnquery_prefix)
        self.order_keys = []                                        # processing orders registered during this pass (dropped with it)
        self.order_refs = []                                        # ... the tensors themselves, alive as long as this cache is
        self.transpose_keys = []                                    # transposed neighbour tables registered during this pass (dropped with it)

    def hint(self, xyz, nsample, algo="set", new_xyz=None, offset=None, new_offset=None):
        """Declare that this geometry will be searched with up to `nsample` neighbours during the pass (a network knows its config:
        the blocks' K = 8 / 16 and the CBL head's K = 36 look at the same points).  The first narrower request then runs the WIDE search
        once (tie policy `algo`) and every narrower one is derived from it — rows decided by a tie are replayed, so the values are those
        of the separate searches.  Without offsets the hint applies to any offsets used with these coordinates."""
        self.hints[self._geo(xyz, xyz if new_xyz is None else new_xyz)] = (int(nsample), algo)

    def _geo(self, xyz, new_xyz):
        return (xyz.data_ptr(), tuple(xyz.shape), new_xyz.data_ptr(), tuple(new_xyz.shape))

    def __enter__(self):
        self._prev = neighbor_cache.active()
        neighbor_cache._tls.cache = self
        return self

    def __exit__(self, *exc):
        neighbor_cache._tls.cache = self._prev
        if not getattr(self, "keep", False):
            self.store.clear()
            self.host.clear()
            self.wide.clear()
            for key in self.order_keys:
                _order_registry.pop(key, None)
            self.order_keys.clear()
            self.order_refs.clear()
            for key in self.transpose_keys:
                _transpose_registry.pop(key, None)
            self.transpose_keys.clear()
        return False

    @staticmethod
    def active():
        return getattr(neighbor_cache._tls, "cache", None)

    ignore_version = False               # static geometry (geometry.StaticGeometry): tensors are refreshed IN PLACE between uses

    def _key(self, kind, algo, tensors):
        if self.ignore_version:
            return (kind, algo) + tuple((t.data_ptr(), tuple(t.shape)) for t in tensors)
        return (kind, algo) + tuple((t.data_ptr(), tuple(t.shape), _version_of(t)) for t in tensors)

    def _host_key(self, o):
        return (o.data_ptr(), tuple(o.shape)) if self.ignore_version else (o.data_ptr(), tuple(o.shape), _version_of(o))

    # Entries may have been produced on ANOTHER stream (geometry prefetch, contrastboundary_amd/geometry.py): each carries the
    # event recorded behind its producer; a consumer stream waits for it and is registered with the allocator as a user.
    def _deliver(self, entry):
        outs, _keys, event, stream = entry
        if event is not None:
            cur = torch.cuda.current_stream(outs[0].device)
            if stream != cur:
                cur.wait_event(event)
                for t in outs:
                    t.record_stream(cur)
        return outs

    def _stamp(self, outs, keys):
        dev = outs[0].device
        if getattr(self, "record_events", False):
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            return (outs, keys, ev, torch.cuda.current_stream(dev))
        return (outs, keys, None, None)

    def lookup(self, nsample, algo, tensors):
        for a in ((algo, "auto") if algo == "set" else (algo,)):
            hit = self.store.get(self._key(nsample, a, tensors))
            if hit is not None:
                self.hits += 1
                return self._deliver(hit)
        self.misses += 1
        return None

    def insert(self, nsample, algo, tensors, idx, dist2, event=None):
        if event is not None:                                        # recorded by the producer itself (behind the part that made idx / dist2)
            self.store[self._key(nsample, algo, tensors)] = ((idx, dist2), tensors, event, torch.cuda.current_stream(idx.device))
        else:
            self.store[self._key(nsample, algo, tensors)] = self._stamp((idx, dist2), tensors)
        geo = self._geo(tensors[0], tensors[1])
        if algo in ("auto", "set", "grid") and nsample > self.wide.get(geo, (0, None))[0]:
            self.wide[geo] = (nsample, algo)

    def wider(self, nsample, tensors):
        """-> (nsample_wide, idx_wide, dist2_wide) of a stored wider result over the same tensors, or None"""
        w = self.wide.get(self._geo(tensors[0], tensors[1]))
        if w is None or w[0] <= nsample:
            return None
        hit = self.store.get(self._key(w[0], w[1], tensors))
        if hit is None:
            return None
        idx, dist2 = self._deliver(hit)
        return w[0], idx, dist2

    def lookup_fps(self, stride, tensors):
        hit = self.store.get(self._key(("fps", stride), "", tensors))
        return None if hit is None else self._deliver(hit)

    def insert_fps(self, stride, tensors, new_p, new_o, idx):
        self.store[self._key(("fps", stride), "", tensors)] = self._stamp((new_p, new_o, idx), tensors)


def host_offsets(o):
    """cumulative end offsets `o` (b) as a python list.  Inside a neighbour cache the answer is remembered per tensor (the cache keeps
    the tensor alive, so its storage cannot be recycled under the key) and offsets made by `fps_downsample` are known without asking
    the device at all; outside a cache this is the same blocking read as the reference's `offset[i].item()` loops."""
    cache = neighbor_cache.active()
    if cache is None:
        return o.cpu().tolist()
    key = cache._host_key(o)
    hit = cache.host.get(key)
    if hit is None:
        hit = cache.host[key] = (o.cpu().tolist(), o)
    return hit[0]


