"""dense.linear (streaming kernels for the Linear layers inside the vector attention, csrc/skinny_linear.hip) against
torch.nn.functional.linear in float64: forward, input / weight / bias gradients, for the widths the network uses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cin,cout,bias", [(163840, 3, 3, True), (163840, 3, 64, True), (163840, 64, 8, True), (163840, 8, 8, True),
                                                (40960, 128, 16, True), (327680, 3, 32, True), (327680, 32, 4, False), (20001, 7, 5, True),
                                                (9000, 64, 64, True),
                                                # widths that are multiples of 16 (<= 64) run on the matrix cores: q / k / v of the two full-resolution stages
                                                (40960, 64, 64, True), (40961, 32, 32, True), (10243, 16, 48, False), (8200, 48, 16, True), (9999, 64, 32, True),
                                                # ragged c_in beside a tiled c_out (TransitionDown: Linear(3 + C, C')): operands padded in registers
                                                (163840, 35, 64, False), (40963, 35, 64, True), (9001, 19, 32, True), (12345, 50, 16, False), (8193, 63, 48, True)])
def test_skinny_linear_matches_torch(rows, cin, cout, bias):
    from contrastboundary_amd import dense
    torch.manual_seed(rows % 97 + cin)
    x = torch.randn(rows, cin, device="cuda", requires_grad=True)
    w = (torch.randn(cout, cin, device="cuda") / cin ** 0.5).requires_grad_(True)
    b = torch.randn(cout, device="cuda", requires_grad=True) if bias else None
    g = torch.randn(rows, cout, device="cuda")
    assert dense._fits(rows, cin, cout)
    y = dense.linear(x, w, b)
    y.backward(g)
    x64, w64, g64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True), g.double()
    b64 = b.detach().double().requires_grad_(True) if bias else None
    r = torch.nn.functional.linear(x64, w64, b64)
    r.backward(g64)
    close = lambda a, ref, tol: float((a.double() - ref).abs().max()) <= tol * max(float(ref.abs().max()), 1e-30)
    assert close(y, r, 1e-5)
    assert close(x.grad, x64.grad, 1e-5)
    assert close(w.grad, w64.grad, 2e-5)                     # a sum over `rows` terms in fp32
    if bias:
        assert close(b.grad, b64.grad, 2e-5)


@pytest.mark.parametrize("shape,cin,cout,bias", [((2560, 16), 67, 128, False), ((20000,), 131, 256, True), ((16384,), 128, 128, False), ((40961,), 259, 512, True)])
def test_tall_linear_splits_its_weight_gradient(shape, cin, cout, bias):
    """dense.linear outside the streaming kernels' range with many rows (dense._TallLinear: the library's forward and input gradient, the weight gradient as a
    batched product over row chunks) against F.linear in float64"""
    from contrastboundary_amd import dense
    torch.manual_seed(cin)
    x = torch.randn(*shape, cin, device="cuda", requires_grad=True)
    w = (torch.randn(cout, cin, device="cuda") / cin ** 0.5).requires_grad_(True)
    b = torch.randn(cout, device="cuda", requires_grad=True) if bias else None
    g = torch.randn(*shape, cout, device="cuda")
    rows = x.numel() // cin
    assert not dense._fits(rows, cin, cout) and rows >= dense.TALL_ROWS
    y = dense.linear(x, w, b)
    y.backward(g)
    x64, w64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    b64 = b.detach().double().requires_grad_(True) if bias else None
    r = torch.nn.functional.linear(x64, w64, b64)
    r.backward(g.double())
    close = lambda a, ref, tol: float((a.double() - ref).abs().max()) <= tol * max(float(ref.abs().max()), 1e-30)
    assert close(y, r, 2e-5) and close(x.grad, x64.grad, 2e-5) and close(w.grad, w64.grad, 5e-5)
    if bias:
        assert close(b.grad, b64.grad, 5e-5)


def test_linear_falls_back_to_torch_outside_its_range():
    from contrastboundary_amd import dense
    x = torch.randn(100, 64, device="cuda"); w = torch.randn(512, 64, device="cuda")
    assert not dense._fits(100, 64, 512)
    assert torch.equal(dense.linear(x, w), torch.nn.functional.linear(x, w))
    x3 = torch.randn(50, 16, 3, device="cuda"); w3 = torch.randn(3, 3, device="cuda"); b3 = torch.randn(3, device="cuda")
    assert dense.linear(x3, w3, b3).shape == (50, 16, 3)


@pytest.mark.parametrize("shape,relu", [((163840, 64), True), ((163840, 3), True), ((163840, 8), False), ((40960, 32), True),
                                        ((10240, 16, 64), True), ((5000, 6), False), ((20000, 512), True), ((4096, 100), True),
                                        # rows <= 4096: one kernel per direction (the coarse stages of the network)
                                        ((2560, 128), True), ((640, 256), False), ((160, 512), True), ((4096, 32), True), ((100, 16), True), ((257, 4), False),
                                        ((3000, 6), True)])
def test_batch_norm_rows_matches_torch(shape, relu):
    """dense.batch_norm (csrc/bn_rows.hip) against nn.BatchNorm1d (+ ReLU) in float64: output, input / affine gradients, running statistics"""
    import copy
    from contrastboundary_amd import dense
    torch.manual_seed(shape[0] % 89)
    C = shape[-1]
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    ref = copy.deepcopy(bn).double()
    x = (torch.randn(*shape, device="cuda") * 1.7 + 0.3).requires_grad_(True)
    g = torch.randn(*shape, device="cuda")
    y = dense.batch_norm(x, bn, relu=relu)
    y.backward(g)
    x64 = x.detach().double().requires_grad_(True)
    r = ref(x64.reshape(-1, C)).view(shape)
    if relu:
        r = torch.relu(r)
    r.backward(g.double())
    close = lambda a, b, tol: float((a.double() - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-30)
    assert close(y.detach(), r.detach(), 2e-5)
    # a ReLU decision may flip where |y| ~ 1e-7: compare the gradients in L2
    l2 = lambda a, b: float((a.double() - b).norm() / b.norm())
    assert l2(x.grad, x64.grad) < 1e-4
    assert l2(bn.weight.grad, ref.weight.grad) < 1e-4 and l2(bn.bias.grad, ref.bias.grad) < 1e-4
    assert close(bn.running_mean, ref.running_mean, 1e-5) and close(bn.running_var, ref.running_var, 1e-5)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1


@pytest.mark.parametrize("shape,relu", [((163840, 32), True), ((40960, 64), True), ((10240, 64), False), ((2560, 128), True), ((640, 256), True), ((160, 512), False),
                                        ((5000, 6), True), ((3000, 6), True), ((100, 16), True)])
def test_batch_norm_with_residual_matches_torch(shape, relu):
    """[relu](bn(x) + residual) as one call each way (cbl_bn_rows_*_residual: the tail of a residual block, blocks.py:130-133) against nn.BatchNorm1d, `+` and ReLU in
    float64: output, the gradients of x, of the residual and of the affine parameters, running statistics"""
    import copy
    from contrastboundary_amd import dense
    torch.manual_seed(shape[0] % 97)
    C = shape[-1]
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    ref = copy.deepcopy(bn).double()
    x = (torch.randn(*shape, device="cuda") * 1.7 + 0.3).requires_grad_(True)
    res = torch.randn(*shape, device="cuda").requires_grad_(True)
    g = torch.randn(*shape, device="cuda")
    y = dense.batch_norm(x, bn, relu=relu, residual=res)
    y.backward(g)
    x64 = x.detach().double().requires_grad_(True); r64 = res.detach().double().requires_grad_(True)
    r = ref(x64.reshape(-1, C)).view(shape) + r64
    if relu:
        r = torch.relu(r)
    r.backward(g.double())
    close = lambda a, b, tol: float((a.double() - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-30)
    assert close(y.detach(), r.detach(), 2e-5)
    l2 = lambda a, b: float((a.double() - b).norm() / b.norm())
    assert l2(x.grad, x64.grad) < 1e-4 and l2(res.grad, r64.grad) < 1e-4
    assert l2(bn.weight.grad, ref.weight.grad) < 1e-4 and l2(bn.bias.grad, ref.bias.grad) < 1e-4
    assert close(bn.running_mean, ref.running_mean, 1e-5) and close(bn.running_var, ref.running_var, 1e-5)
    # the residual that needs no gradient (a block input without one) and the same tensor on both sides (x + x)
    y2 = dense.batch_norm(x.detach().requires_grad_(True), bn, relu=relu, residual=res.detach())
    assert close(y2.detach(), r.detach(), 2e-5)


def test_batch_norm_eval_mode_and_small_inputs_use_torch():
    from contrastboundary_amd import dense
    bn = torch.nn.BatchNorm1d(16).cuda().eval()
    x = torch.randn(10000, 16, device="cuda")
    assert torch.equal(dense.batch_norm(x, bn, relu=True), torch.relu(bn(x)))
    bn.train()
    xs = torch.randn(40, 16, device="cuda")
    bn2 = torch.nn.BatchNorm1d(16).cuda().train()
    assert torch.allclose(dense.batch_norm(xs, bn, relu=False), bn2(xs), atol=1e-6)


@pytest.mark.parametrize("rows,C", [(40960, 64), (10000, 32), (8193, 64)])
def test_triple_linear_equals_three_linear_layers(rows, C):
    """cbl_triple_linear_* (blocks.py:33: the q / k / v projections, one launch per direction) against torch's F.linear in float64"""
    import torch.nn as nn
    from contrastboundary_amd import dense
    torch.manual_seed(rows)
    ls = [nn.Linear(C, C).cuda() for _ in range(3)]
    x = torch.randn(rows, C, device="cuda", requires_grad=True)
    gs = [torch.randn(rows, C, device="cuda") for _ in range(3)]
    ys = dense.triple_linear(x, *ls)
    torch.autograd.backward(ys, gs)
    got = [y.detach() for y in ys], x.grad.clone(), [l.weight.grad.clone() for l in ls], [l.bias.grad.clone() for l in ls]
    x64 = x.detach().double().requires_grad_(True)
    ws = [l.weight.detach().double().requires_grad_(True) for l in ls]; bs = [l.bias.detach().double().requires_grad_(True) for l in ls]
    ys64 = [torch.nn.functional.linear(x64, w, b) for w, b in zip(ws, bs)]
    torch.autograd.backward(ys64, [g.double() for g in gs])
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    for y, y64 in zip(got[0], ys64):
        assert rel(y, y64.detach()) < 1e-6
    assert rel(got[1], x64.grad) < 1e-6
    for gw, w in zip(got[2], ws):
        assert rel(gw, w.grad) < 1e-5
    for gb, b in zip(got[3], bs):
        assert rel(gb, b.grad) < 1e-5
