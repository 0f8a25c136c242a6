"""Round-2 GPU probe: transposed table build, gather-form backward passes, the atomic-free CBL gradient, against round 1's kernels."""
import ctypes
import numpy as np, torch
from contrastboundary_amd import _lib, pointops, hotpath, heads, neighbor_state, local_aggregation as LA

def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

L = _lib.lib()
_i = ctypes.c_int
n, K, C = 40960, 16, 64
sc = hotpath.Scene.synthetic(n, C, seed=0)
idx, _ = pointops.knnquery_raw(K, sc.xyz, sc.xyz, sc.offset, sc.offset)
widx, _ = pointops.knnquery_raw(36, sc.xyz, sc.xyz, sc.offset, sc.offset, algo="set")
order = pointops.spatial_order(idx)
print("order available:", order is not None)

def transpose(ix, ordr):
    m, ns = ix.shape
    inv_start = torch.empty(n + 1, dtype=torch.int32, device="cuda"); inv_src = torch.empty(m * ns, dtype=torch.int32, device="cuda")
    ws = torch.empty(L.cbl_neighbor_transpose_workspace_bytes(_i(m), _i(n), _i(ns)), dtype=torch.uint8, device="cuda")
    def run():
        _lib.check(L.cbl_neighbor_transpose(_i(m), _i(n), _i(ns), _lib.ptr(ix), _lib.ptr(ordr), _lib.ptr(ordr), _lib.ptr(inv_start), _lib.ptr(inv_src),
                                            _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(ix)), "t")
    return run, inv_start, inv_src

for name, ix in (("K=16", idx), ("K=36", widx)):
    for oname, o in (("cell order", order), ("no order", None)):
        run, s, src = transpose(ix, o)
        print(f"neighbor_transpose {name} {oname:10s}: {timeit(run):8.1f} us")

# grouping backward: atomics vs gather
go = torch.randn(n, K, C, device="cuda")
gi = torch.zeros(n, C, device="cuda")
def atomic():
    gi.zero_()
    _lib.check(L.cbl_grouping_backward(_i(n), _i(K), _i(C), _lib.ptr(go), _lib.ptr(idx), _lib.ptr(gi), _lib.stream_of(go)), "a")
t_at = timeit(atomic)
ref = gi.clone()
run, s16, src16 = transpose(idx, order); run(); torch.cuda.synchronize()
gi2 = torch.empty(n, C, device="cuda")
def csr():
    _lib.check(L.cbl_grouping_backward_csr(_i(n), _i(C), _lib.ptr(go), _lib.ptr(order), _lib.ptr(s16), _lib.ptr(src16), _lib.ptr(gi2), _lib.stream_of(go)), "c")
t_csr = timeit(csr)
byt = 4*n*K + 4*n*K*C + 4*n*C
print(f"grouping bwd atomics (+zero fill): {t_at:8.1f} us  {byt/t_at/1e3:7.1f} GB/s")
print(f"grouping bwd gather (CSR)        : {t_csr:8.1f} us  {byt/t_csr/1e3:7.1f} GB/s  max|diff| vs atomics {float((gi2-ref).abs().max()):.2e}")
run_n, s16n, src16n = transpose(idx, None); run_n(); torch.cuda.synchronize()
def csr_noorder():
    _lib.check(L.cbl_grouping_backward_csr(_i(n), _i(C), _lib.ptr(go), None, _lib.ptr(s16n), _lib.ptr(src16n), _lib.ptr(gi2), _lib.stream_of(go)), "c")
print(f"grouping bwd gather, no order    : {timeit(csr_noorder):8.1f} us")
gow = torch.randn(n, K, 3 + C, device="cuda")
def csr_rows():
    _lib.check(L.cbl_grouping_backward_csr_rows(_i(n), _i(C), _i(3 + C), _i(3), _lib.ptr(gow), _lib.ptr(order), _lib.ptr(s16), _lib.ptr(src16), _lib.ptr(gi2), _lib.stream_of(go)), "r")
t_rows = timeit(csr_rows)
print(f"grouping bwd gather, 3+C rows    : {t_rows:8.1f} us  {byt/t_rows/1e3:7.1f} GB/s")

# CBL: round 1 fused (atomics) vs pairs + gather
d = 32
f = sc.latent
amax = sc.labels.to(torch.int32)
per_point = torch.empty(n, device="cuda"); mask = torch.empty(n, dtype=torch.int32, device="cuda")
stats = torch.empty(2, device="cuda"); loss = torch.empty(1, device="cuda"); unit = torch.zeros_like(f)
def old():
    unit.zero_()
    _lib.check(L.cbl_point_contrast_forward_grad(_i(n), _i(36), _i(d), _lib.ptr(f), _lib.ptr(amax), _lib.ptr(widx), ctypes.c_float(1.0), ctypes.c_float(0.1),
                                                 _lib.ptr(per_point), _lib.ptr(mask), _lib.ptr(stats), _lib.ptr(loss), _lib.ptr(unit), _lib.stream_of(f)), "o")
print(f"CBL round-1 fused fwd+grad (atomics, + zero fill + finalize): {timeit(old):8.1f} us   valid points {int(mask.sum())}")
worder = pointops.spatial_order(widx)
run36, s36, src36 = transpose(widx, worder); run36(); torch.cuda.synchronize()
coef = torch.empty(n, 36, device="cuda"); own = torch.empty(n, d, device="cuda"); g = torch.empty_like(f); gl = torch.ones(1, device="cuda")
for oname, o in (("cell order", worder), ("no order", None)):
    if o is None:
        r2, s36b, src36b = transpose(widx, None); r2(); torch.cuda.synchronize()
    else:
        s36b, src36b = s36, src36
    def fwd():
        _lib.check(L.cbl_contrast_pairs_forward(_i(n), _i(0x7fffffff), _i(0), _i(36), _i(d), _lib.ptr(f), _lib.ptr(amax), _i(0), ctypes.c_float(0), _lib.ptr(widx),
                                                _lib.ptr(o), ctypes.c_float(1.0), ctypes.c_float(0.1), _lib.ptr(per_point), _lib.ptr(mask), _lib.ptr(stats),
                                                _lib.ptr(loss), _lib.ptr(coef), _lib.ptr(own), _lib.stream_of(f)), "f")
    def fwd_nograd():
        _lib.check(L.cbl_contrast_pairs_forward(_i(n), _i(0x7fffffff), _i(0), _i(36), _i(d), _lib.ptr(f), _lib.ptr(amax), _i(0), ctypes.c_float(0), _lib.ptr(widx),
                                                _lib.ptr(o), ctypes.c_float(1.0), ctypes.c_float(0.1), _lib.ptr(per_point), _lib.ptr(mask), _lib.ptr(stats),
                                                _lib.ptr(loss), None, None, _lib.stream_of(f)), "f")
    def bwd():
        _lib.check(L.cbl_contrast_pairs_backward(_i(n), _i(36), _i(d), _lib.ptr(f), _lib.ptr(coef), _lib.ptr(own), _lib.ptr(o), _lib.ptr(s36b), _lib.ptr(src36b),
                                                 _lib.ptr(stats), _lib.ptr(gl), ctypes.c_float(0.1), _lib.ptr(g), _lib.stream_of(f)), "b")
    print(f"CBL pairs forward (+finalize) {oname:10s}: {timeit(fwd):8.1f} us   forward only: {timeit(fwd_nograd):8.1f} us   gather backward: {timeit(bwd):8.1f} us")
fwd(); bwd(); old(); torch.cuda.synchronize()
oldg = unit * (0.1 / stats[1])
print("CBL grad max|new-old| / max|old| :", float((g - oldg).abs().max() / oldg.abs().max()))

# KPConv
f64 = sc.feat.clone().requires_grad_(True); kw = sc.kernel_weights.clone().requires_grad_(True); gk = torch.randn(n, C, device="cuda")
out = LA.kpconv(sc.xyz, sc.xyz, idx, f64, sc.kernel_points, kw, 0.12)
def kb():
    f64.grad = None; kw.grad = None
    out.backward(gk, retain_graph=True)
print(f"kpconv fwd: {timeit(lambda: LA.kpconv(sc.xyz, sc.xyz, idx, sc.feat, sc.kernel_points, sc.kernel_weights, 0.12)):8.1f} us   kpconv bwd (atomics, incl. zero fills): {timeit(kb):8.1f} us")

# events recorded inside a captured graph
try:
    evs = [torch.cuda.Event(enable_timing=True, external=True) for _ in range(3)]
    x = torch.zeros(1 << 24, device="cuda")
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        x.add_(1)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        evs[0].record(); x.add_(1); evs[1].record(); x.add_(1); x.add_(1); evs[2].record()
    for _ in range(3):
        gph.replay(); torch.cuda.synchronize()
        print("in-graph events:", evs[0].elapsed_time(evs[1]), evs[1].elapsed_time(evs[2]))
except Exception as e:
    print("in-graph events FAILED:", type(e).__name__, e)
