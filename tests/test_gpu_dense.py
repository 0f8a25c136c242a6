"""dense.linear (streaming kernels for the Linear layers inside the vector attention, csrc/skinny_linear.hip) against
torch.nn.functional.linear in float64: forward, input / weight / bias gradients, for the widths the network uses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cin,cout,bias", [(163840, 3, 3, True), (163840, 3, 64, True), (163840, 64, 8, True), (163840, 8, 8, True),
                                                (40960, 128, 16, True), (327680, 3, 32, True), (327680, 32, 4, False), (20001, 7, 5, True),
                                                (9000, 64, 64, True)])
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


def test_linear_falls_back_to_torch_outside_its_range():
    from contrastboundary_amd import dense
    x = torch.randn(100, 64, device="cuda"); w = torch.randn(512, 64, device="cuda")
    assert not dense._fits(100, 64, 512)
    assert torch.equal(dense.linear(x, w), torch.nn.functional.linear(x, w))
    x3 = torch.randn(50, 16, 3, device="cuda"); w3 = torch.randn(3, 3, device="cuda"); b3 = torch.randn(3, device="cuda")
    assert dense.linear(x3, w3, b3).shape == (50, 16, 3)
