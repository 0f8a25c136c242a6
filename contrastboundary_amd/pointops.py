"""Host-side mirror of the reference's pointops Python API, over libcbl_amd.so (hand-written HIP, gfx950).

Same names, argument order and value semantics as /root/reference/pytorch/lib/pointops/functions/pointops.py:
    furthestsampling :27   knnquery :45   grouping :76   queryandgroup :79   subtraction :130
    aggregation :161       interpolation :164             interpolation2 :214
Tensors are stacked clouds `(sum n_i, .)` with int32 cumulative end `offset (b,)`.  torch is used for device
memory, streams and autograd plumbing only; every op body is a C-ABI call (include/cbl_amd.h).
Unlike the reference (no checks at all), dtype / device / contiguity are validated and a
failing launch raises instead of surfacing later as an asynchronous error.
"""
import ctypes

import torch
from torch.autograd import Function

from . import _lib

_c_int = ctypes.c_int


def _req(t, dtype, name, dim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise _lib.CblError(f"{name}: must live on the GPU (got {t.device}); there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if dim is not None and t.dim() != dim:
        raise ValueError(f"{name}: expected {dim} dims, got shape {tuple(t.shape)}")
    return t


def _as_int(v):
    """nsample / k may arrive as a 0-dim tensor (heads.py:190, basic_operators.py:22)"""
    return int(v.item()) if isinstance(v, torch.Tensor) else int(v)


from .neighbor_state import (ORDER_MIN_POINTS, _ORDER_REGISTRY_MAX, _order_alias, _order_key, _order_register, _order_registry,  # noqa: F401
                             _order_wanted, _workspace, _ws_cache, host_offsets, neighbor_cache, spatial_order)
from . import neighbor_state


# ------------------------------------------------------------------------------------------------ K2
def _furthestsampling_raw(xyz, offset, new_offset, n_max, m, cert_in=None, want_cert=False):
    """the launch alone: n_max (longest cloud) and m (= new_offset[-1]) come from the host side, nothing synchronises.
    want_cert / cert_in: cbl_furthestsampling_chain (-> idx, cert): sampling the samples of an earlier run is its prefix where that run certified it"""
    n, b = xyz.shape[0], offset.shape[0]
    idx = torch.zeros(m, dtype=torch.int32, device=xyz.device)
    tmp = torch.full((n,), 1e10, dtype=torch.float32, device=xyz.device)
    L = _lib.lib()
    need = L.cbl_furthestsampling_workspace_bytes(_c_int(b), _c_int(n), _c_int(n_max))     # > 0: large clouds, bucket-pruned kernel
    ws = _workspace(need, xyz.device)
    if want_cert or cert_in is not None:
        cert = torch.empty(b, dtype=torch.int32, device=xyz.device)
        rc = L.cbl_furthestsampling_chain(_c_int(b), _c_int(n), _c_int(n_max), _lib.ptr(xyz), _lib.ptr(offset), _lib.ptr(new_offset), _lib.ptr(tmp),
                                          _lib.ptr(idx), _lib.ptr(cert_in), _lib.ptr(cert), _lib.ptr(ws),
                                          ctypes.c_size_t(ws.numel() if ws is not None else 0), _lib.stream_of(xyz))
        _lib.check(rc, "cbl_furthestsampling_chain")
        return idx, cert
    rc = L.cbl_furthestsampling_ws(_c_int(b), _c_int(n), _c_int(n_max), _lib.ptr(xyz), _lib.ptr(offset), _lib.ptr(new_offset),
                                   _lib.ptr(tmp), _lib.ptr(idx), _lib.ptr(ws), ctypes.c_size_t(ws.numel() if ws is not None else 0),
                                   _lib.stream_of(xyz))
    _lib.check(rc, "cbl_furthestsampling_ws")
    return idx


class FurthestSampling(Function):
    @staticmethod
    def forward(ctx, xyz, offset, new_offset):
        """xyz (n,3) f32, offset (b) i32, new_offset (b) i32 -> idx (m) i32          pointops.py:12-25"""
        _req(xyz, torch.float32, "xyz", 2); _req(offset, torch.int32, "offset", 1); _req(new_offset, torch.int32, "new_offset", 1)
        b = offset.shape[0]
        off_h = offset.cpu()                       # the reference also syncs here (n_max, new_offset[b-1].item())
        lens = torch.diff(off_h, prepend=off_h.new_zeros(1))
        n_max = int(lens.max().item()) if b > 0 else 0
        m = int(new_offset[b - 1].item()) if b > 0 else 0
        return _furthestsampling_raw(xyz, offset, new_offset, n_max, m)


furthestsampling = FurthestSampling.apply

# ------------------------------------------------------------------------------------------------ K1
_tie_policy = "reference"


def set_knn_tie_policy(policy):
    """'reference' (default): neighbour order and choice under exactly equal distances as the reference's heap produces them, bit for bit
    (tied queries are replayed through that heap: costly on quantised coordinates).  'set': exact set, free order among equal
    distances.  'anytie': exact distances, free choice among supports tied at the K-th distance; cost independent of ties.  The
    reference's networks are invariant to both freedoms (softmax / max / mean over the K set).  Returns the previous policy."""
    global _tie_policy
    if policy not in ("reference", "set", "anytie"):
        raise ValueError(policy)
    prev, _tie_policy = _tie_policy, policy
    return prev


fps_prefix_chain = True      # False: every stage runs the sampler (measurement / tests)


def fps_downsample(p, o, stride):
    """TransitionDown's sampling step (blocks.py:61-68): per cloud n_b // stride furthest-point samples.
    -> (new_p (m,3), new_o (b) i32, idx (m) i32); cached per forward like the neighbour searches (the coordinates handed back are the
    SAME tensor on a hit, so later searches on them hit the cache as well).  Inside a cache no step of it waits for the device."""
    cache = neighbor_cache.active()
    if cache is not None:
        hit = cache.lookup_fps(stride, (p, o))
        if hit is not None:
            return hit
    ends = host_offsets(o)
    lens = [e - s for s, e in zip([0] + ends[:-1], ends)]
    new_ends, run = [], 0
    for l in lens:
        run += l // stride
        new_ends.append(run)
    staged = torch.tensor(new_ends, dtype=torch.int32).pin_memory()
    new_o = staged.to(o.device, non_blocking=True)
    # the samples of an FPS run, sampled again, are that run's prefix where it certified its arg-maxima as unique (cbl_furthestsampling_chain): the
    # certificate rides on the sampled tensor itself (same object, same version, same number of clouds), so only a genuine chain can use it.
    # (`_version` counts torch's in-place operations only: a caller that rewrites the sampled coordinates through a raw pointer — the C ABI, another
    # library — must drop the attribute itself; nothing in this package writes a sampled tensor in place.)
    tag = getattr(p, "_fps_certificate", None)
    cert_in = tag[0] if (fps_prefix_chain and tag is not None and tag[1] == p._version and tag[2] is o and tag[3] == o._version) else None
    idx, cert = _furthestsampling_raw(p, o, new_o, max(lens) if lens else 0, run, cert_in=cert_in, want_cert=True)
    new_p = p[idx.long(), :]
    new_p._fps_certificate = (cert, new_p._version, new_o, new_o._version)   # valid for exactly these rows and these cloud boundaries
    if cache is not None:
        cache.host[cache._host_key(new_o)] = (new_ends, new_o, staged)
        cache.insert_fps(stride, (p, o), new_p, new_o, idx)
    return new_p, new_o, idx


def knnquery_raw(nsample, xyz, new_xyz, offset, new_offset, algo="auto"):
    """-> idx (m,nsample) i32, dist2 (m,nsample) f32 (squared).  algo: 'auto' | 'exact' | 'grid' | 'set' | 'anytie'
    ('set': same neighbour set and distances, order among exactly equal distances unspecified — cbl_knnquery_set;
     'anytie': same distances, any of the supports tied at the K-th distance — cbl_knnquery_anytie, never replays).
    `set_knn_tie_policy` redirects 'auto' requests of the mirrors (blocks, interpolation, heads) to one of the cheaper policies."""
    nsample = _as_int(nsample)
    if new_xyz is None:
        new_xyz = xyz
    # validated once, before the cache logic: the nested / derived paths below reach the kernels without passing _knnquery_uncached
    _req(xyz, torch.float32, "xyz", 2); _req(new_xyz, torch.float32, "new_xyz", 2)
    _req(offset, torch.int32, "offset", 1); _req(new_offset, torch.int32, "new_offset", 1)
    if not 1 <= nsample <= 1024:
        raise ValueError(f"nsample={nsample} outside [1, 1024] (knnquery_cuda_kernel.cu:89)")
    if xyz.shape[1] != 3 or new_xyz.shape[1] != 3:
        raise ValueError("xyz / new_xyz: expected (n, 3)")
    if algo == "auto" and _tie_policy != "reference":
        algo = _tie_policy
    cache = neighbor_cache.active()
    if cache is not None:
        tensors = (xyz, new_xyz, offset, new_offset)
        hit = cache.lookup(nsample, algo, tensors)
        if hit is not None:
            return hit
        if algo in ("auto", "set"):
            # a narrower search over a geometry that has (or, by a hint, will need) a wider one: derive it (cbl_knnquery_prefix)
            wide = cache.wider(nsample, tensors)
            if wide is None:
                h = cache.hints.get(cache._geo(xyz, new_xyz))
                if h is not None and h[0] > nsample and h[0] <= 64:
                    # one C call for both; with cross-stream consumers the event behind the WIDE search alone is recorded inside the call
                    ev = None
                    if getattr(cache, "record_events", False):
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream(xyz.device))        # creates the handle; re-recorded inside the call
                    both = _knnquery_nested(h[0], h[1], nsample, algo, xyz, new_xyz, offset, new_offset, ev)
                    if both is not None:                             # wide search + derivation in one C call
                        wi, wd, idx, dist2 = both
                        cache.insert(h[0], h[1], tensors, wi, wd, event=ev)
                        cache.insert(nsample, algo, tensors, idx, dist2)
                        cache.derived += 1
                        return idx, dist2
                    wi, wd = _knnquery_uncached(h[0], xyz, new_xyz, offset, new_offset, h[1])
                    cache.insert(h[0], h[1], tensors, wi, wd)
                    wide = (h[0], wi, wd)
            if wide is not None:
                idx, dist2 = knn_prefix(nsample, wide[0], wide[1], wide[2], xyz, new_xyz, offset, new_offset, algo)
                cache.insert(nsample, algo, tensors, idx, dist2)
                cache.derived += 1
                if new_xyz is xyz or (new_xyz.data_ptr() == xyz.data_ptr() and new_xyz.shape == xyz.shape):
                    _order_alias(idx, xyz)
                return idx, dist2
        idx, dist2 = _knnquery_uncached(nsample, xyz, new_xyz, offset, new_offset, algo)
        cache.insert(nsample, algo, tensors, idx, dist2)
        return idx, dist2
    return _knnquery_uncached(nsample, xyz, new_xyz, offset, new_offset, algo)


def knn_prefix(nsample, nsample_wide, idx_wide, dist2_wide, xyz, new_xyz, offset, new_offset, algo="auto"):
    """the `nsample` nearest neighbours from a wider result over the same tensors (cbl_knnquery_prefix): same values as
    knnquery_raw(nsample, ..., algo) with algo 'auto' (reference order) or 'set'"""
    if algo not in ("auto", "set"):
        raise ValueError("knn_prefix: algo must be 'auto' or 'set'")
    n, m, b = xyz.shape[0], new_xyz.shape[0], offset.shape[0]
    idx = torch.empty((m, nsample), dtype=torch.int32, device=xyz.device)
    dist2 = torch.empty((m, nsample), dtype=torch.float32, device=xyz.device)
    L = _lib.lib()
    ws = _workspace(max(L.cbl_knnquery_prefix_workspace_bytes(_c_int(m)), 1), xyz.device)
    _lib.check(L.cbl_knnquery_prefix(_c_int(b), _c_int(n), _c_int(m), _c_int(nsample_wide), _c_int(nsample), _lib.ptr(xyz), _lib.ptr(new_xyz),
                                     _lib.ptr(offset), _lib.ptr(new_offset), _lib.ptr(idx_wide), _lib.ptr(dist2_wide), _lib.ptr(idx), _lib.ptr(dist2),
                                     _c_int(1 if algo == "set" else 0), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(xyz)),
               "cbl_knnquery_prefix")
    return idx, dist2


def _knnquery_nested(nsample_wide, algo_wide, nsample, algo, xyz, new_xyz, offset, new_offset, event=None):
    """cbl_knnquery_nested: -> (idx_wide, dist2_wide, idx, dist2), or None where the wide search would not take the grid path"""
    if algo_wide not in ("auto", "set", "anytie", "grid") or algo not in ("auto", "set"):
        return None
    n, m, b = xyz.shape[0], new_xyz.shape[0], offset.shape[0]
    L = _lib.lib()
    need = L.cbl_knnquery_workspace_bytes(_c_int(b), _c_int(n), _c_int(m), _c_int(nsample_wide))
    if need == 0 or nsample_wide > 64:
        return None
    dev = xyz.device
    wi = torch.empty((m, nsample_wide), dtype=torch.int32, device=dev); wd = torch.empty((m, nsample_wide), dtype=torch.float32, device=dev)
    idx = torch.empty((m, nsample), dtype=torch.int32, device=dev); dist2 = torch.empty((m, nsample), dtype=torch.float32, device=dev)
    ws = _workspace(need, dev)
    cur = torch.cuda.current_stream(dev)
    self_search = new_xyz is xyz or (new_xyz.data_ptr() == xyz.data_ptr() and m == n)
    order = torch.empty(n, dtype=torch.int32, device=dev) if (self_search and _order_wanted(xyz, cur.cuda_stream)) else None
    pw = 1 if algo_wide == "set" else 2 if algo_wide == "anytie" else 0
    rc = L.cbl_knnquery_nested(_c_int(b), _c_int(n), _c_int(m), _c_int(nsample_wide), _c_int(pw), _c_int(nsample), _c_int(1 if algo == "set" else 0),
                               _lib.ptr(xyz), _lib.ptr(new_xyz), _lib.ptr(offset), _lib.ptr(new_offset), _lib.ptr(wi), _lib.ptr(wd), _lib.ptr(idx),
                               _lib.ptr(dist2), _lib.ptr(order), ctypes.c_void_p(event.cuda_event if event is not None else 0),
                               _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(xyz))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "cbl_knnquery_nested")
    if order is not None:
        _order_register(xyz, order, cur)
    if self_search and n >= ORDER_MIN_POINTS and neighbor_state.use_spatial_order:
        _order_alias(wi, xyz); _order_alias(idx, xyz)
    return wi, wd, idx, dist2


def _knnquery_uncached(nsample, xyz, new_xyz, offset, new_offset, algo):
    n, m, b = xyz.shape[0], new_xyz.shape[0], offset.shape[0]
    # every element is written by the kernels, so no zero fill (the reference zero-fills, pointops.py:40-41)
    idx = torch.empty((m, nsample), dtype=torch.int32, device=xyz.device)
    dist2 = torch.empty((m, nsample), dtype=torch.float32, device=xyz.device)
    L = _lib.lib()
    st = _lib.stream_of(xyz)
    args = (_c_int(b), _c_int(n), _c_int(m), _c_int(nsample), _lib.ptr(xyz), _lib.ptr(new_xyz), _lib.ptr(offset),
            _lib.ptr(new_offset), _lib.ptr(idx), _lib.ptr(dist2))
    if algo == "exact":
        _lib.check(L.cbl_knnquery_exact(*args, st), "cbl_knnquery_exact")
    else:
        need = L.cbl_knnquery_workspace_bytes(_c_int(b), _c_int(n), _c_int(m), _c_int(nsample))
        if algo == "grid" and need == 0:
            raise _lib.CblError("grid KNN not available for this problem shape")
        ws = _workspace(need, xyz.device)
        cur = torch.cuda.current_stream(xyz.device)
        self_search = new_xyz is xyz or (new_xyz.data_ptr() == xyz.data_ptr() and m == n)
        if self_search and need and nsample <= 64 and _order_wanted(xyz, cur.cuda_stream):
            # the grid build lists the supports cell by cell anyway: keep that sequence as the processing order of this geometry
            order = torch.empty(n, dtype=torch.int32, device=xyz.device)
            policy = 1 if algo == "set" else 2 if algo == "anytie" else 0
            rc = L.cbl_knnquery_ordered(*args, _c_int(policy), _lib.ptr(order), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), st)
            if rc == 0:
                _order_register(xyz, order, cur)
                _order_alias(idx, xyz)
                return idx, dist2
            if rc != _lib.ERR_UNSUPPORTED:
                _lib.check(rc, "cbl_knnquery_ordered")
        fn = L.cbl_knnquery_set if algo == "set" else L.cbl_knnquery_anytie if algo == "anytie" else L.cbl_knnquery
        _lib.check(fn(*args, _lib.ptr(ws), ctypes.c_size_t(ws.numel() if ws is not None else 0), st), "cbl_knnquery")
        if self_search and n >= ORDER_MIN_POINTS and neighbor_state.use_spatial_order:
            _order_alias(idx, xyz)
    return idx, dist2


def knn_block_candidates(nsample, xyz, offset, algo="set"):
    """measurement support (bench.py `roofline.search`): run the grid self-search over `xyz` and return, per query, how many candidate supports its
    27-cell block held — the pairs the search evaluated in its first round ("pairs visited") -> int32 (n,).  CblError off the grid path."""
    n, b = xyz.shape[0], offset.shape[0]
    L = _lib.lib()
    need = L.cbl_knnquery_workspace_bytes(_c_int(b), _c_int(n), _c_int(n), _c_int(nsample))
    if need == 0:
        raise _lib.CblError("knn_block_candidates: this shape does not take the grid search")
    _knnquery_uncached(nsample, xyz, xyz, offset, offset, algo)
    ws = _workspace(need, xyz.device)                               # the same buffer (one per device and stream), holding the grid that search built
    count = torch.empty(n, dtype=torch.int32, device=xyz.device)
    _lib.check(L.cbl_knn_grid_block_candidates(_c_int(b), _c_int(n), _c_int(nsample), _lib.ptr(offset), _lib.ptr(count), _lib.ptr(ws),
                                               ctypes.c_size_t(ws.numel()), _lib.stream_of(xyz)), "cbl_knn_grid_block_candidates")
    return count


# ------------------------------------------------------------------------------------------------ transposed neighbour table
TRANSPOSE_MIN_PAIRS = 1 << 16      # below this a scatter with atomics is as quick as building the table (unless the table is cached)


def neighbor_transpose(idx, n, build=True, companion=None):
    """Transposed neighbour table of idx (m, nsample) over n target rows (cbl_neighbor_transpose, SURVEY.md 7 hard part 6):
    -> (order or None, inv_start (n+1) i32, inv_src (m*nsample) i32), or None (build=False and nothing cached).
    Segment r of inv_src lists, ascending, the flat pairs p = source * nsample + column with idx[p] == order[r] (== r without an order).
    It depends on the table alone: it is built once per table (kept in neighbor_state's registry; a neighbour cache drops what was built
    during its pass) and shared by every backward pass that scatters through it, also on autograd's thread; the processing order (cell
    order of the search that made idx) only makes build and consumers local.
    companion: another neighbour table (m, nsample') of the SAME geometry (same m sources, n targets — the block's K = 8 / 16 table beside the CBL head's
    K = 36 table of a stage) whose transposed table a later pass of the step will ask for: both are built by the same four launches
    (cbl_neighbor_transpose_pair) and registered; the later request is a registry hit on whatever stream it comes from."""
    _req(idx, torch.int32, "idx", 2)
    hit = neighbor_state.transpose_lookup(idx)
    if hit is not None or not build:
        return hit
    m, nsample = idx.shape
    order = spatial_order(idx) if m == n else None
    L = _lib.lib()
    inv_start = torch.empty(n + 1, dtype=torch.int32, device=idx.device)
    inv_src = torch.empty(max(m * nsample, 1), dtype=torch.int32, device=idx.device)
    if companion is not None and companion.dim() == 2 and companion.shape[0] == m and companion.dtype == torch.int32 and companion.device == idx.device \
            and companion.data_ptr() != idx.data_ptr() and neighbor_state.transpose_lookup(companion) is None:
        ns2 = companion.shape[1]
        inv_start2 = torch.empty(n + 1, dtype=torch.int32, device=idx.device)
        inv_src2 = torch.empty(max(m * ns2, 1), dtype=torch.int32, device=idx.device)
        ws = _workspace(max(L.cbl_neighbor_transpose_pair_workspace_bytes(_c_int(m), _c_int(n), _c_int(nsample), _c_int(ns2)), 1), idx.device)
        rc = L.cbl_neighbor_transpose_pair(_c_int(m), _c_int(n), _c_int(nsample), _lib.ptr(idx), _c_int(ns2), _lib.ptr(companion), _lib.ptr(order), _lib.ptr(order),
                                           _lib.ptr(inv_start), _lib.ptr(inv_src), _lib.ptr(inv_start2), _lib.ptr(inv_src2), _lib.ptr(ws), ctypes.c_size_t(ws.numel()),
                                           _lib.stream_of(idx))
        if rc == _lib.ERR_UNSUPPORTED:
            return None
        _lib.check(rc, "cbl_neighbor_transpose_pair")
        neighbor_state.transpose_register(idx, order, inv_start, inv_src)
        neighbor_state.transpose_register(companion, order, inv_start2, inv_src2)
        return order, inv_start, inv_src
    ws = _workspace(max(L.cbl_neighbor_transpose_workspace_bytes(_c_int(m), _c_int(n), _c_int(nsample)), 1), idx.device)
    rc = L.cbl_neighbor_transpose(_c_int(m), _c_int(n), _c_int(nsample), _lib.ptr(idx), _lib.ptr(order), _lib.ptr(order), _lib.ptr(inv_start),
                                  _lib.ptr(inv_src), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(idx))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "cbl_neighbor_transpose")
    neighbor_state.transpose_register(idx, order, inv_start, inv_src)
    return order, inv_start, inv_src


def _scatter_rows(grad_rows, idx, n, col0=0, c=None):
    """grad_in (n, c) = scatter-add of grad_rows[..., col0 : col0 + c] (grad_rows (m, nsample, width) contiguous) through idx: a gather over
    the transposed table where that pays (or the table is already there), the reference's atomic scatter (grouping_cuda_kernel.cu:16-25) otherwise"""
    m, nsample, width = grad_rows.shape
    c = width - col0 if c is None else c
    L = _lib.lib()
    tr = neighbor_transpose(idx, n, build=(m * nsample >= TRANSPOSE_MIN_PAIRS))
    if tr is not None:
        order, inv_start, inv_src = tr
        grad_in = torch.empty((n, c), dtype=torch.float32, device=grad_rows.device)
        _lib.check(L.cbl_grouping_backward_csr_rows(_c_int(n), _c_int(c), _c_int(width), _c_int(col0), _lib.ptr(grad_rows), _lib.ptr(order),
                                                    _lib.ptr(inv_start), _lib.ptr(inv_src), _lib.ptr(grad_in), _lib.stream_of(grad_rows)),
                   "cbl_grouping_backward_csr_rows")
        return grad_in
    if col0 or c != width:
        grad_rows = grad_rows[..., col0:col0 + c].contiguous()
    grad_in = torch.zeros((n, c), dtype=torch.float32, device=grad_rows.device)
    _lib.check(L.cbl_grouping_backward(_c_int(m), _c_int(nsample), _c_int(c), _lib.ptr(grad_rows), _lib.ptr(idx), _lib.ptr(grad_in),
                                       _lib.stream_of(grad_rows)), "cbl_grouping_backward")
    return grad_in


class KNNQuery(Function):
    @staticmethod
    def forward(ctx, nsample, xyz, new_xyz, offset, new_offset):
        """-> idx (m,nsample) i32, dist (m,nsample) f32 = sqrt(dist2)                   pointops.py:32-43"""
        idx, dist2 = knnquery_raw(nsample, xyz, new_xyz, offset, new_offset)
        if neighbor_cache.active() is not None:
            idx = idx.view(idx.shape)          # a cached result is shared by several calls: hand autograd its own tensor object
        ctx.mark_non_differentiable(idx)
        return idx, torch.sqrt(dist2)

    @staticmethod
    def backward(ctx, *grads):      # indices are not differentiable; the reference defines no backward
        return None, None, None, None, None


knnquery = KNNQuery.apply


def knn_indices(nsample, xyz, new_xyz, offset, new_offset):
    """the neighbour table alone, for the mirrors' own calls that drop knnquery's second output (blocks.py:34-35, pointops.py:88-89 `idx, _ = knnquery(...)`): the
    same search, cache and tensor as `knnquery(...)[0]`, without the sqrt launch over distances nobody reads (22 of them per training step of the network)"""
    idx, _ = knnquery_raw(nsample, xyz, new_xyz if new_xyz is not None else xyz, offset, new_offset)
    if neighbor_cache.active() is not None:
        idx = idx.view(idx.shape)              # a cached result is shared by several calls: every caller its own tensor object (as KNNQuery.forward)
    return idx


# ------------------------------------------------------------------------------------------------ K3/K4
class Grouping(Function):
    @staticmethod
    def forward(ctx, input, idx):
        """input (n,c), idx (m,nsample) -> (m,nsample,c)                               pointops.py:50-61"""
        _req(input, torch.float32, "input", 2); _req(idx, torch.int32, "idx", 2)
        m, nsample, n, c = idx.shape[0], idx.shape[1], input.shape[0], input.shape[1]
        output = torch.empty((m, nsample, c), dtype=torch.float32, device=input.device)
        order = spatial_order(idx)                               # processing order only: same values (cbl_amd.h)
        _lib.check(_lib.lib().cbl_grouping_forward_ordered(_c_int(m), _c_int(nsample), _c_int(c), _lib.ptr(input), _lib.ptr(idx), _lib.ptr(order),
                                                           _lib.ptr(output), _lib.stream_of(input)), "cbl_grouping_forward")
        ctx.n = n
        ctx.save_for_backward(idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        idx, = ctx.saved_tensors
        grad_output = grad_output.contiguous()          # the reference forgets this
        return _scatter_rows(grad_output, idx, ctx.n), None


grouping = Grouping.apply


# ------------------------------------------------------------------------------------------------ F1
class _QueryAndGroup(Function):
    """fused gather(xyz)-centre + gather(feat) + concat of pointops.py:90-98; backward = scatter-add"""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feat, idx, use_xyz):
        m, nsample = idx.shape
        c = feat.shape[1]
        oc = c + (3 if use_xyz else 0)
        out = torch.empty((m, nsample, oc), dtype=torch.float32, device=feat.device)
        order = spatial_order(new_xyz) if use_xyz else None      # processing order only: same values (cbl_amd.h)
        _lib.check(_lib.lib().cbl_queryandgroup_ordered(_c_int(m), _c_int(nsample), _c_int(c), _c_int(1 if use_xyz else 0), _lib.ptr(xyz),
                                                        _lib.ptr(new_xyz), _lib.ptr(feat), _lib.ptr(idx), _lib.ptr(order), _lib.ptr(out),
                                                        _lib.stream_of(feat)), "cbl_queryandgroup")
        ctx.save_for_backward(idx)
        ctx.dims = (xyz.shape[0], feat.shape[0], c, use_xyz)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, = ctx.saved_tensors
        n_xyz, n_feat, c, use_xyz = ctx.dims
        m, nsample = idx.shape
        L = _lib.lib()
        g_xyz = g_new = g_feat = None
        grad_out = grad_out.contiguous()
        if ctx.needs_input_grad[2]:
            g_feat = _scatter_rows(grad_out, idx, n_feat, 3 if use_xyz else 0, c)      # the feature columns, read where they lie
        if use_xyz and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            if ctx.needs_input_grad[0]:
                g_xyz = _scatter_rows(grad_out, idx, n_xyz, 0, 3)
            if ctx.needs_input_grad[1]:
                g_new = -grad_out[..., :3].sum(1)
        return g_xyz, g_new, g_feat, None, None


def queryandgroup(nsample, xyz, new_xyz, feat, idx, offset, new_offset, use_xyz=True):
    """-> (m,nsample,3+c) if use_xyz else (m,nsample,c)                               pointops.py:79-100"""
    if new_xyz is None:
        new_xyz = xyz
    _req(xyz, torch.float32, "xyz", 2); _req(new_xyz, torch.float32, "new_xyz", 2); _req(feat, torch.float32, "feat", 2)
    if idx is None:
        idx = knn_indices(nsample, xyz, new_xyz, offset, new_offset)
    _req(idx, torch.int32, "idx", 2)
    return _QueryAndGroup.apply(xyz, new_xyz, feat, idx, bool(use_xyz))


# ------------------------------------------------------------------------------------------------ K7/K8
class Subtraction(Function):
    @staticmethod
    def forward(ctx, input1, input2, idx):
        """input1 (n,c), input2 (n,c), idx (n,nsample) -> (n,nsample,c)               pointops.py:103-115"""
        _req(input1, torch.float32, "input1", 2); _req(input2, torch.float32, "input2", 2); _req(idx, torch.int32, "idx", 2)
        n, c = input1.shape
        nsample = idx.shape[-1]
        output = torch.empty((n, nsample, c), dtype=torch.float32, device=input1.device)
        _lib.check(_lib.lib().cbl_subtraction_forward_ordered(_c_int(n), _c_int(nsample), _c_int(c), _lib.ptr(input1), _lib.ptr(input2),
                                                              _lib.ptr(idx), _lib.ptr(spatial_order(idx)), _lib.ptr(output), _lib.stream_of(input1)),
                   "cbl_subtraction_forward")
        ctx.save_for_backward(idx)
        ctx.n2 = input2.shape[0]
        return output

    @staticmethod
    def backward(ctx, grad_output):
        idx, = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, nsample, c = grad_output.shape
        g1 = torch.zeros((n, c), dtype=torch.float32, device=grad_output.device)
        tr = neighbor_transpose(idx, ctx.n2, build=(n * nsample >= TRANSPOSE_MIN_PAIRS))
        if tr is not None:                                           # K8 as a gather over the transposed table: no atomics (cbl_amd.h)
            order, inv_start, inv_src = tr
            g2 = torch.empty((ctx.n2, c), dtype=torch.float32, device=grad_output.device)
            _lib.check(_lib.lib().cbl_subtraction_backward_csr(_c_int(n), _c_int(ctx.n2), _c_int(nsample), _c_int(c), _lib.ptr(grad_output), _lib.ptr(order),
                                                               _lib.ptr(inv_start), _lib.ptr(inv_src), _lib.ptr(g1), _lib.ptr(g2),
                                                               _lib.stream_of(grad_output)), "cbl_subtraction_backward_csr")
            return g1, g2, None
        g2 = torch.zeros((ctx.n2, c), dtype=torch.float32, device=grad_output.device)
        _lib.check(_lib.lib().cbl_subtraction_backward(_c_int(n), _c_int(nsample), _c_int(c), _lib.ptr(idx), _lib.ptr(grad_output),
                                                       _lib.ptr(g1), _lib.ptr(g2), _lib.stream_of(grad_output)), "cbl_subtraction_backward")
        return g1, g2, None


subtraction = Subtraction.apply


# ------------------------------------------------------------------------------------------------ K9/K10
class Aggregation(Function):
    @staticmethod
    def forward(ctx, input, position, weight, idx):
        """input (n,c), position (n,nsample,c), weight (n,nsample,c'), idx (n,nsample) -> (n,c)   pointops.py:133-144"""
        _req(input, torch.float32, "input", 2); _req(position, torch.float32, "position", 3)
        _req(weight, torch.float32, "weight", 3); _req(idx, torch.int32, "idx", 2)
        n, nsample, c = position.shape
        w_c = weight.shape[-1]
        output = torch.zeros((n, c), dtype=torch.float32, device=input.device)
        _lib.check(_lib.lib().cbl_aggregation_forward_ordered(_c_int(n), _c_int(nsample), _c_int(c), _c_int(w_c), _lib.ptr(input), _lib.ptr(position),
                                                              _lib.ptr(weight), _lib.ptr(idx), _lib.ptr(spatial_order(idx)), _lib.ptr(output),
                                                              _lib.stream_of(input)), "cbl_aggregation_forward")
        ctx.save_for_backward(input, position, weight, idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, position, weight, idx = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, nsample, c = position.shape
        w_c = weight.shape[-1]
        dev = grad_output.device
        gp = torch.zeros((n, nsample, c), dtype=torch.float32, device=dev)
        gw = torch.zeros((n, nsample, w_c), dtype=torch.float32, device=dev)
        tr = neighbor_transpose(idx, input.shape[0], build=(n * nsample >= TRANSPOSE_MIN_PAIRS))
        if tr is not None:                                           # grad_input of K10 as a gather over the transposed table, the per-pair outputs as before
            order, inv_start, inv_src = tr
            L = _lib.lib()
            gi = torch.empty((input.shape[0], c), dtype=torch.float32, device=dev)
            _lib.check(L.cbl_aggregation_backward(_c_int(n), _c_int(nsample), _c_int(c), _c_int(w_c), _lib.ptr(input), _lib.ptr(position), _lib.ptr(weight),
                                                  _lib.ptr(idx), _lib.ptr(grad_output), None, _lib.ptr(gp), _lib.ptr(gw), _lib.stream_of(grad_output)),
                       "cbl_aggregation_backward")
            _lib.check(L.cbl_weighted_scatter_csr(_c_int(input.shape[0]), _c_int(nsample), _c_int(c), _c_int(w_c), _lib.ptr(grad_output), _lib.ptr(weight),
                                                  _lib.ptr(order), _lib.ptr(inv_start), _lib.ptr(inv_src), _lib.ptr(gi), _lib.stream_of(grad_output)),
                       "cbl_weighted_scatter_csr")
            return gi, gp, gw, None
        gi = torch.zeros((input.shape[0], c), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().cbl_aggregation_backward(_c_int(n), _c_int(nsample), _c_int(c), _c_int(w_c), _lib.ptr(input), _lib.ptr(position),
                                                       _lib.ptr(weight), _lib.ptr(idx), _lib.ptr(grad_output), _lib.ptr(gi), _lib.ptr(gp),
                                                       _lib.ptr(gw), _lib.stream_of(grad_output)), "cbl_aggregation_backward")
        return gi, gp, gw, None


aggregation = Aggregation.apply


# ------------------------------------------------------------------------------------------------ F4 / K5 / K6
def _interp_idx_weight(xyz, new_xyz, offset, new_offset, k):
    idx, dist2 = knnquery_raw(k, xyz, new_xyz, offset, new_offset)
    n = new_xyz.shape[0]
    weight = torch.empty((n, k), dtype=torch.float32, device=xyz.device)
    _lib.check(_lib.lib().cbl_interpolation_weights(_c_int(n), _c_int(k), _lib.ptr(dist2), _lib.ptr(weight), ctypes.c_void_p(0),
                                                    _lib.stream_of(xyz)), "cbl_interpolation_weights")
    return idx, weight


class Interpolation(Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, input, offset, new_offset, k=3):
        """xyz (m,3) coarse, new_xyz (n,3) fine, input (m,c) -> (n,c)                  pointops.py:181-199"""
        k = _as_int(k)
        _req(input, torch.float32, "input", 2)
        idx, weight = _interp_idx_weight(xyz, new_xyz, offset, new_offset, k)
        n, c, m = new_xyz.shape[0], input.shape[1], input.shape[0]
        output = torch.zeros((n, c), dtype=torch.float32, device=input.device)
        _lib.check(_lib.lib().cbl_interpolation_forward(_c_int(n), _c_int(c), _c_int(k), _lib.ptr(input), _lib.ptr(idx), _lib.ptr(weight),
                                                        _lib.ptr(output), _lib.stream_of(input)), "cbl_interpolation_forward")
        ctx.m, ctx.k = m, k
        ctx.save_for_backward(idx, weight)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        idx, weight = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, c = grad_output.shape
        tr = neighbor_transpose(idx, ctx.m, build=(n * ctx.k >= TRANSPOSE_MIN_PAIRS))
        if tr is not None:                                           # K6 as a gather over the transposed table: no atomics (cbl_amd.h)
            order, inv_start, inv_src = tr
            grad_input = torch.empty((ctx.m, c), dtype=torch.float32, device=grad_output.device)
            _lib.check(_lib.lib().cbl_weighted_scatter_csr(_c_int(ctx.m), _c_int(ctx.k), _c_int(c), _c_int(1), _lib.ptr(grad_output), _lib.ptr(weight),
                                                           _lib.ptr(order), _lib.ptr(inv_start), _lib.ptr(inv_src), _lib.ptr(grad_input),
                                                           _lib.stream_of(grad_output)), "cbl_weighted_scatter_csr")
            return None, None, grad_input, None, None, None
        grad_input = torch.zeros((ctx.m, c), dtype=torch.float32, device=grad_output.device)
        _lib.check(_lib.lib().cbl_interpolation_backward(_c_int(n), _c_int(c), _c_int(ctx.k), _lib.ptr(grad_output), _lib.ptr(idx),
                                                         _lib.ptr(weight), _lib.ptr(grad_input), _lib.stream_of(grad_output)),
                   "cbl_interpolation_backward")
        return None, None, grad_input, None, None, None


interpolation2 = Interpolation.apply


class WeightedGather(Function):
    """output[i] = sum_k weight[i, k] input[idx[i, k]] for a GIVEN neighbour table — K5 / K6 (interpolation_cuda_kernel.cu) behind a table the caller
    already has (basic_operators.get_subscene_features: the mean over the kr nearest stage-0 points, weight = 1 / kr)"""

    @staticmethod
    def forward(ctx, input, idx, weight):
        _req(input, torch.float32, "input", 2); _req(idx, torch.int32, "idx", 2); _req(weight, torch.float32, "weight", 2)
        n, k = idx.shape
        c, m = input.shape[1], input.shape[0]
        output = torch.zeros((n, c), dtype=torch.float32, device=input.device)
        _lib.check(_lib.lib().cbl_interpolation_forward(_c_int(n), _c_int(c), _c_int(k), _lib.ptr(input), _lib.ptr(idx), _lib.ptr(weight),
                                                        _lib.ptr(output), _lib.stream_of(input)), "cbl_interpolation_forward")
        ctx.m, ctx.k = m, k
        ctx.save_for_backward(idx, weight)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        idx, weight = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        n, c = grad_output.shape
        grad_input = torch.zeros((ctx.m, c), dtype=torch.float32, device=grad_output.device)
        _lib.check(_lib.lib().cbl_interpolation_backward(_c_int(n), _c_int(c), _c_int(ctx.k), _lib.ptr(grad_output), _lib.ptr(idx),
                                                         _lib.ptr(weight), _lib.ptr(grad_input), _lib.stream_of(grad_output)),
                   "cbl_interpolation_backward")
        return grad_input, None, None


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """pure-torch composite in the reference (pointops.py:164-178); same values, one gather kernel here"""
    return Interpolation.apply(xyz, new_xyz, feat, offset, new_offset, k)
