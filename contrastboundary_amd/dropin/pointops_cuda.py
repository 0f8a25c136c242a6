"""Drop-in replacement for the reference's CUDA extension module `pointops_cuda`
(/root/reference/pytorch/lib/pointops/src/pointops_api.cpp:12-23): put this directory on sys.path and the reference's
pytorch/lib/pointops/functions/pointops.py (`import pointops_cuda`, :7) runs unmodified on PyTorch-ROCm.  Same 10 function
names and positional signatures; bodies are C-ABI calls into libcbl_amd.so (include/cbl_amd.h).  See INTEGRATION.md §1."""
# torch's wheel carries its own libamdhip64 and the process must hold a single HIP runtime.
import ctypes, torch
import os
_L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib", "libcbl_amd.so"))
_L.cbl_knnquery_workspace_bytes.restype = ctypes.c_size_t
_L.cbl_furthestsampling_workspace_bytes.restype = ctypes.c_size_t
_s = lambda t: ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
def _ok(rc, what):
    if rc: raise RuntimeError(f"{what}: error {rc}")
def _chk(t, dtype, name, shape=None):
    """the reference's launchers check nothing (SURVEY 8(b) "Error convention"): a wrong dtype / device / layout is silent garbage there, an error here"""
    if not isinstance(t, torch.Tensor): raise TypeError(f"pointops_cuda: {name} must be a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda: raise RuntimeError(f"pointops_cuda: {name} must be a CUDA tensor (got {t.device}); there is no CPU path")
    if t.dtype != dtype: raise TypeError(f"pointops_cuda: {name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous(): raise ValueError(f"pointops_cuda: {name} must be contiguous (shape {tuple(t.shape)}, stride {t.stride()})")
    if shape is not None and tuple(t.shape) != tuple(shape): raise ValueError(f"pointops_cuda: {name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return ctypes.c_void_p(t.data_ptr())
_f = lambda t, name, shape=None: _chk(t, torch.float32, name, shape)
_i = lambda t, name, shape=None: _chk(t, torch.int32, name, shape)
_n = lambda v: int(v)                                                  # scalars may arrive as 0-dim tensors (heads.py:186 nsample[i], pointops.py:18 n_max)
_ws = {}
def _scratch(need, t):
    if not need: return None, 0
    key = (t.device, torch.cuda.current_stream(t.device).cuda_stream)  # one scratch per stream: two streams never share a workspace
    ws = _ws.get(key)
    if ws is None or ws.numel() < need: ws = _ws[key] = torch.empty(need, dtype=torch.uint8, device=t.device)
    return ctypes.c_void_p(ws.data_ptr()), ws.numel()

def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):          # pointops_api.cpp:13
    m, nsample = _n(m), _n(nsample)
    b, n = offset.shape[0], xyz.shape[0]
    args = (_f(xyz, "xyz", (n, 3)), _f(new_xyz, "new_xyz", (m, 3)), _i(offset, "offset", (b,)), _i(new_offset, "new_offset", (b,)),
            _i(idx, "idx", (m, nsample)), _f(dist2, "dist2", (m, nsample)))
    ws, nws = _scratch(_L.cbl_knnquery_workspace_bytes(b, n, m, nsample), xyz)
    _ok(_L.cbl_knnquery(b, n, m, nsample, *args, ws, ctypes.c_size_t(nws), _s(xyz)), "cbl_knnquery")

def furthestsampling_cuda(b, n_max, xyz, offset, new_offset, tmp, idx):                # :14  (n_max may be a 0-dim tensor)
    b, n_max = _n(b), _n(n_max)
    n = xyz.shape[0]
    args = (_f(xyz, "xyz", (n, 3)), _i(offset, "offset", (b,)), _i(new_offset, "new_offset", (b,)), _f(tmp, "tmp", (n,)), _i(idx, "idx"))
    ws, nws = _scratch(_L.cbl_furthestsampling_workspace_bytes(b, n, n_max), xyz)     # > 0: large clouds take the bucket-pruned kernel
    _ok(_L.cbl_furthestsampling_ws(b, n, n_max, *args, ws, ctypes.c_size_t(nws), _s(xyz)), "cbl_furthestsampling_ws")

def grouping_forward_cuda(m, nsample, c, input, idx, output):                          # :15
    m, nsample, c = _n(m), _n(nsample), _n(c)
    _ok(_L.cbl_grouping_forward(m, nsample, c, _f(input, "input"), _i(idx, "idx", (m, nsample)), _f(output, "output", (m, nsample, c)), _s(input)), "cbl_grouping_forward")
def grouping_backward_cuda(m, nsample, c, grad_output, idx, grad_input):               # :16
    m, nsample, c = _n(m), _n(nsample), _n(c)
    g = grad_output.contiguous()                                                       # the reference forgets this (pointops.py:64-74)
    _ok(_L.cbl_grouping_backward(m, nsample, c, _f(g, "grad_output", (m, nsample, c)), _i(idx, "idx", (m, nsample)), _f(grad_input, "grad_input"), _s(g)), "cbl_grouping_backward")
def interpolation_forward_cuda(n, c, k, input, idx, weight, output):                   # :17
    n, c, k = _n(n), _n(c), _n(k)
    _ok(_L.cbl_interpolation_forward(n, c, k, _f(input, "input"), _i(idx, "idx", (n, k)), _f(weight, "weight", (n, k)), _f(output, "output", (n, c)), _s(input)), "cbl_interpolation_forward")
def interpolation_backward_cuda(n, c, k, grad_output, idx, weight, grad_input):        # :18
    n, c, k = _n(n), _n(c), _n(k)
    g = grad_output.contiguous()
    _ok(_L.cbl_interpolation_backward(n, c, k, _f(g, "grad_output", (n, c)), _i(idx, "idx", (n, k)), _f(weight, "weight", (n, k)), _f(grad_input, "grad_input"), _s(g)), "cbl_interpolation_backward")
def subtraction_forward_cuda(n, nsample, c, input1, input2, idx, output):              # :19
    n, nsample, c = _n(n), _n(nsample), _n(c)
    _ok(_L.cbl_subtraction_forward(n, nsample, c, _f(input1, "input1", (n, c)), _f(input2, "input2"), _i(idx, "idx", (n, nsample)), _f(output, "output", (n, nsample, c)), _s(input1)), "cbl_subtraction_forward")
def subtraction_backward_cuda(n, nsample, c, idx, grad_output, grad_input1, grad_input2):   # :20
    n, nsample, c = _n(n), _n(nsample), _n(c)
    g = grad_output.contiguous()
    _ok(_L.cbl_subtraction_backward(n, nsample, c, _i(idx, "idx", (n, nsample)), _f(g, "grad_output", (n, nsample, c)), _f(grad_input1, "grad_input1", (n, c)), _f(grad_input2, "grad_input2"), _s(g)), "cbl_subtraction_backward")
def aggregation_forward_cuda(n, nsample, c, w_c, input, position, weight, idx, output):     # :21
    n, nsample, c, w_c = _n(n), _n(nsample), _n(c), _n(w_c)
    _ok(_L.cbl_aggregation_forward(n, nsample, c, w_c, _f(input, "input"), _f(position, "position", (n, nsample, c)), _f(weight, "weight", (n, nsample, w_c)),
                                   _i(idx, "idx", (n, nsample)), _f(output, "output", (n, c)), _s(input)), "cbl_aggregation_forward")
def aggregation_backward_cuda(n, nsample, c, w_c, input, position, weight, idx, grad_output, grad_input, grad_position, grad_weight):   # :22
    n, nsample, c, w_c = _n(n), _n(nsample), _n(c), _n(w_c)
    g = grad_output.contiguous()
    _ok(_L.cbl_aggregation_backward(n, nsample, c, w_c, _f(input, "input"), _f(position, "position", (n, nsample, c)), _f(weight, "weight", (n, nsample, w_c)),
                                    _i(idx, "idx", (n, nsample)), _f(g, "grad_output", (n, c)), _f(grad_input, "grad_input"),
                                    _f(grad_position, "grad_position", (n, nsample, c)), _f(grad_weight, "grad_weight", (n, nsample, w_c)), _s(g)), "cbl_aggregation_backward")
