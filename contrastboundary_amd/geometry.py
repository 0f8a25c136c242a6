"""Geometry prefetch for the Point Transformer + CBL network (one pyramid call, GPU dataloader stage).

Everything the network needs from the coordinates alone — the furthest-point samples of the four down-sampling stages and every
neighbour search of the forward pass and of the criterion — depends on nothing the network computes.  `prefetch` issues all of it
on a SIDE stream into a `pointops.neighbor_cache`; a forward pass run inside that cache (`with geom: ...`) finds every request
answered and only waits, per request, for the event behind its producer.  Used one batch ahead (the data loader knows the next
batch), the ~13 ms of sequential FPS latency of a 40960-point scene — one workgroup on one CU — disappear behind the previous
step's dense compute on the other 255 CUs.

The requests replayed here are exactly those of blocks.py / heads.py / basic_operators.py (same functions, same arguments), so a
cache miss is impossible to distinguish from a hit by its result; the tests run the network both ways and compare bitwise.
"""
import torch

from . import pointops

_side_streams = {}


def side_stream(device):
    """the device's geometry stream: one that really runs beside the caller's (hotpath.concurrent_streams observes which fresh streams share a hardware queue
    with it; during a graph capture nothing can be observed and a plain stream is taken)"""
    s = _side_streams.get(device)
    if s is None:
        if torch.cuda.is_current_stream_capturing():
            return torch.cuda.Stream(device=device)
        from . import hotpath
        with torch.cuda.device(device):
            s = _side_streams[device] = hotpath.concurrent_streams(1, beside=[torch.cuda.current_stream(device)])[0]
    return s


def prefetch(points, offset, stride=(1, 4, 4, 4, 4), nsample=(8, 16, 16, 16, 16), cbl_nsample=None, nstride=None, multi_head=True, stream=None,
             cache=None, after_caller=True):
    """points (n,3) f32, offset (b) i32 on the GPU -> a `pointops.neighbor_cache` being filled on `stream` (default: the device's side
    stream).  cbl_nsample / nstride: the criterion's config.nsample / config.nstride (None: no CBL searches)."""
    dev = points.device
    stream = stream if stream is not None else side_stream(dev)
    if cache is None:
        cache = pointops.neighbor_cache()
        cache.record_events = True
    cache.keep = True                    # entries outlive the `with` used to fill them
    if after_caller:
        stream.wait_stream(torch.cuda.current_stream(dev))          # the inputs were produced on the caller's stream
    with torch.cuda.stream(stream), torch.no_grad(), cache:
        p, o = [points], [offset]
        for s in range(1, len(stride)):                             # TransitionDown of stage s (blocks.py:61-69)
            n_p, n_o, _ = pointops.fps_downsample(p[s - 1], o[s - 1], stride[s])
            p.append(n_p); o.append(n_o)
            pointops.knnquery_raw(nsample[s], p[s - 1], n_p, o[s - 1], n_o)                       # queryandgroup of the transition
        for s in range(len(stride)):
            pointops.knnquery_raw(nsample[s], p[s], p[s], o[s], o[s])                             # attention blocks, encoder and decoder
            if s > 0:
                pointops.knnquery_raw(3, p[s], p[s - 1], o[s], o[s - 1])                          # TransitionUp interpolation (blocks.py:108)
                if multi_head:
                    pointops.knnquery_raw(1, p[s], p[0], o[s], o[0])                              # MultiHead.upsample (heads.py:50)
            if cbl_nsample is not None:
                pointops.knnquery_raw(int(cbl_nsample[s]), p[s], p[s], o[s], o[s], algo="set")    # ContrastHead (heads.py:192)
                if s > 0 and nstride is not None:
                    kr = 1
                    for v in nstride[:s]:
                        kr *= int(v)
                    pointops.knnquery_raw(kr, p[0], p[s], o[0], o[s], algo="set")                 # sub-scene labels (basic_operators.py:22-30)
    cache.hits = cache.misses = 0
    return cache


class StaticGeometry:
    """Geometry of one batch in FIXED device tensors, for a training step captured in a hipGraph (torch.cuda.CUDAGraph): the graph bakes
    the addresses of the index / coordinate tensors in, `refresh` recomputes them for the next batch (on a side stream) and copies the
    results over the old ones.  Batches must have the shapes of the first one (same cloud sizes)."""

    def __init__(self, points, offset, **plan):
        self.plan = plan
        self.points, self.offset = points, offset                   # the static input tensors (refreshed in place by the caller)
        self.cache = pointops.neighbor_cache()
        self.cache.ignore_version = True
        prefetch(points, offset, stream=torch.cuda.current_stream(points.device), cache=self.cache, **plan)
        self.ready = None
        # the cloud boundaries on the host, read ONCE: every later batch has the first one's cloud sizes (the class's contract), and refresh() must not ask the
        # device for them — a blocking device-to-host read on the side stream waits for everything queued there, including the event of the replay that last
        # used this buffer set, so the issuing thread sat in stage() for a whole step (measured: 8 - 17 ms of host time per step, the graph step host-bound)
        self.host_ends = offset.cpu().tolist()

    def check_offset(self, offset):
        """a batch about to be staged into this buffer set must have the FIRST batch's cloud boundaries (host_ends: FPS counts, n_max and new_offset were derived
        from them once).  Checked when the boundaries are readable without a device wait (a host / pinned tensor, a list); a device tensor is the caller's promise."""
        if offset is None:
            return
        if torch.is_tensor(offset):
            if offset.is_cuda:
                return
            ends = offset.tolist()
        else:
            ends = [int(v) for v in offset]
        if ends != self.host_ends:
            raise ValueError("StaticGeometry: cloud boundaries %s differ from the captured batch's %s (same total is not enough)" % (ends[:8], self.host_ends[:8]))

    def refresh(self, stream=None):
        """recompute for the CURRENT contents of self.points / self.offset; returns after enqueueing (self.ready = event)"""
        dev = self.points.device
        stream = stream if stream is not None else side_stream(dev)
        fresh = pointops.neighbor_cache()
        fresh.record_events = True
        fresh.host[fresh._host_key(self.offset)] = (self.host_ends, self.offset)                      # known: no device-to-host read in here
        fresh = prefetch(self.points, self.offset, stream=stream, cache=fresh, after_caller=False, **self.plan)   # ordering is the caller's (events)
        with torch.cuda.stream(stream), torch.no_grad():
            old, new = list(self.cache.store.values()), list(fresh.store.values())
            assert len(old) == len(new)
            for (o_outs, *_), (n_outs, *_) in zip(old, new):
                for a, b in zip(o_outs, n_outs):
                    assert a.shape == b.shape, "StaticGeometry: the batch changed shape"
                    a.copy_(b)
            self.ready = torch.cuda.Event()
            self.ready.record(stream)
        # what the scratch pass registered (orders / tables keyed by its temporaries) leaves the registries with it: they are bounded LRUs, and
        # entries piling up here once pushed the STATIC geometry's orders out — freed while captured graphs still read them
        from . import neighbor_state
        for key in fresh.order_keys:
            neighbor_state._order_registry.pop(key, None)
        for key in fresh.transpose_keys:
            neighbor_state._transpose_registry.pop(key, None)
        fresh.order_keys.clear(); fresh.transpose_keys.clear(); fresh.order_refs.clear()
        fresh.store.clear(); fresh.host.clear()
        return self.ready

    def __enter__(self):
        return self.cache.__enter__()

    def __exit__(self, *exc):
        return self.cache.__exit__(*exc)
