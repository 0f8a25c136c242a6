import ctypes, torch
from contrastboundary_amd import _lib, pointops, hotpath
L = _lib.lib(); i = ctypes.c_int
n = 40960
sc = hotpath.Scene.synthetic(n, 64, seed=0)
idx, _ = pointops.knnquery_raw(16, sc.xyz, sc.xyz, sc.offset, sc.offset)
widx, _ = pointops.knnquery_raw(36, sc.xyz, sc.xyz, sc.offset, sc.offset, algo="set")
order = pointops.spatial_order(idx)
for ix in (idx, widx):
    m, ns = ix.shape
    inv_start = torch.empty(n + 1, dtype=torch.int32, device="cuda"); inv_src = torch.empty(m * ns, dtype=torch.int32, device="cuda")
    ws = torch.empty(L.cbl_neighbor_transpose_workspace_bytes(i(m), i(n), i(ns)), dtype=torch.uint8, device="cuda")
    for _ in range(12):
        _lib.check(L.cbl_neighbor_transpose(i(m), i(n), i(ns), _lib.ptr(ix), _lib.ptr(order), _lib.ptr(order), _lib.ptr(inv_start), _lib.ptr(inv_src), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream_of(ix)), "t")
    torch.cuda.synchronize()
