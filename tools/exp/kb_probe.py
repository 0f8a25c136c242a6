"""KPConv backward (gather form): where does the time go?  variants by output set and grid size"""
import ctypes, os, sys
import torch
from contrastboundary_amd import _lib, pointops, hotpath
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
L = _lib.lib(); i = ctypes.c_int
n, K, C, KP = 40960, 16, 64, 15
sc = hotpath.Scene.synthetic(n, C, seed=0)
idx, _ = pointops.knnquery_raw(K, sc.xyz, sc.xyz, sc.offset, sc.offset)
order, s, src = pointops.neighbor_transpose(idx, n)
go = torch.randn(n, C, device="cuda")
gf = torch.empty(n, C, device="cuda"); gkw = torch.empty(KP, C, device="cuda")
ws = torch.empty(L.cbl_kpconv_backward_csr_workspace_bytes(i(n), i(C), i(KP)) * 4, dtype=torch.uint8, device="cuda")
def run(wf, ww, o):
    return lambda: _lib.check(L.cbl_kpconv_backward_csr(i(n), i(n), i(K), i(C), i(KP), _lib.ptr(sc.xyz), _lib.ptr(sc.xyz), _lib.ptr(sc.feat), _lib.ptr(sc.kernel_points),
        _lib.ptr(sc.kernel_weights), ctypes.c_float(0.12), i(1), i(0), _lib.ptr(go), _lib.ptr(o), _lib.ptr(s if o is not None else s_n), _lib.ptr(src if o is not None else src_n),
        _lib.ptr(gf if wf else None), _lib.ptr(gkw if ww else None), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(go)), "kb")
print("grid", "resident", " gf+gkw %.1f us   gf only %.1f us   gkw only %.1f us" % (timeit(run(True, True, order)), timeit(run(True, False, order)), timeit(run(False, True, order))))
deg = (s[1:] - s[:-1]).float()
print("in-degree mean %.1f max %d  >16: %.2f  >32: %.3f" % (deg.mean().item(), int(deg.max()), (deg > 16).float().mean().item(), (deg > 32).float().mean().item()))
