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
_p = lambda t: ctypes.c_void_p(t.data_ptr())
_s = lambda t: ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
def _ok(rc, what):
    if rc: raise RuntimeError(f"{what}: error {rc}")
_ws = {}

def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):          # pointops_api.cpp:13
    b, n = offset.shape[0], xyz.shape[0]
    need = _L.cbl_knnquery_workspace_bytes(b, n, m, nsample)
    ws = _ws.get(xyz.device)
    if need and (ws is None or ws.numel() < need):
        ws = _ws[xyz.device] = torch.empty(need, dtype=torch.uint8, device=xyz.device)
    _ok(_L.cbl_knnquery(b, n, m, nsample, _p(xyz), _p(new_xyz), _p(offset), _p(new_offset), _p(idx), _p(dist2),
                        _p(ws) if need else None, ctypes.c_size_t(ws.numel() if need else 0), _s(xyz)), "cbl_knnquery")

def furthestsampling_cuda(b, n_max, xyz, offset, new_offset, tmp, idx):                # :14  (n_max may be a 0-dim tensor)
    n = xyz.shape[0]
    need = _L.cbl_furthestsampling_workspace_bytes(b, n, int(n_max))                   # > 0: large clouds take the bucket-pruned kernel
    ws = _ws.get(xyz.device)
    if need and (ws is None or ws.numel() < need):
        ws = _ws[xyz.device] = torch.empty(need, dtype=torch.uint8, device=xyz.device)
    _ok(_L.cbl_furthestsampling_ws(b, n, int(n_max), _p(xyz), _p(offset), _p(new_offset), _p(tmp), _p(idx),
                                   _p(ws) if need else None, ctypes.c_size_t(ws.numel() if need else 0), _s(xyz)), "cbl_furthestsampling_ws")

def grouping_forward_cuda(m, nsample, c, input, idx, output):                          # :15
    _ok(_L.cbl_grouping_forward(m, nsample, c, _p(input), _p(idx), _p(output), _s(input)), "cbl_grouping_forward")
def grouping_backward_cuda(m, nsample, c, grad_output, idx, grad_input):               # :16
    g = grad_output.contiguous()
    _ok(_L.cbl_grouping_backward(m, nsample, c, _p(g), _p(idx), _p(grad_input), _s(g)), "cbl_grouping_backward")
def interpolation_forward_cuda(n, c, k, input, idx, weight, output):                   # :17
    _ok(_L.cbl_interpolation_forward(n, c, k, _p(input), _p(idx), _p(weight), _p(output), _s(input)), "cbl_interpolation_forward")
def interpolation_backward_cuda(n, c, k, grad_output, idx, weight, grad_input):        # :18
    g = grad_output.contiguous()
    _ok(_L.cbl_interpolation_backward(n, c, k, _p(g), _p(idx), _p(weight), _p(grad_input), _s(g)), "cbl_interpolation_backward")
def subtraction_forward_cuda(n, nsample, c, input1, input2, idx, output):              # :19
    _ok(_L.cbl_subtraction_forward(n, nsample, c, _p(input1), _p(input2), _p(idx), _p(output), _s(input1)), "cbl_subtraction_forward")
def subtraction_backward_cuda(n, nsample, c, idx, grad_output, grad_input1, grad_input2):   # :20
    g = grad_output.contiguous()
    _ok(_L.cbl_subtraction_backward(n, nsample, c, _p(idx), _p(g), _p(grad_input1), _p(grad_input2), _s(g)), "cbl_subtraction_backward")
def aggregation_forward_cuda(n, nsample, c, w_c, input, position, weight, idx, output):     # :21
    _ok(_L.cbl_aggregation_forward(n, nsample, c, w_c, _p(input), _p(position), _p(weight), _p(idx), _p(output), _s(input)), "cbl_aggregation_forward")
def aggregation_backward_cuda(n, nsample, c, w_c, input, position, weight, idx, grad_output, grad_input, grad_position, grad_weight):   # :22
    g = grad_output.contiguous()
    _ok(_L.cbl_aggregation_backward(n, nsample, c, w_c, _p(input), _p(position), _p(weight), _p(idx), _p(g), _p(grad_input),
                                    _p(grad_position), _p(grad_weight), _s(g)), "cbl_aggregation_backward")
