"""Host logic behind the wide attention layers' batched q / k / v product: the three projections' weights as one (3, C, C) view (pt_layer._stacked) once
pt_layer.adjoin_qkv or distributed.FlatState has laid them back to back; Parameter objects, values and state_dict entries unchanged.  (What the product itself
computes is a GPU test: tests/test_gpu_blocks.py.)"""
import torch

from contrastboundary_amd import blocks, distributed, pt_layer


def _triples(layer):
    return [[getattr(l, name) for l in (layer.linear_q, layer.linear_k, layer.linear_v)] for name in ("weight", "bias")]


def test_adjoin_makes_the_stack_a_view_and_keeps_the_values():
    torch.manual_seed(0)
    layer = blocks.PointTransformerLayer(128, 128, 8, 16)
    before = {k: v.clone() for k, v in layer.state_dict().items()}
    params = [id(p) for p in layer.parameters()]
    for t in _triples(layer):
        assert pt_layer._stacked(t).data_ptr() != t[0].data_ptr()                # separate allocations: a copy
        assert torch.equal(pt_layer._stacked(t), torch.stack(t))
    pt_layer.adjoin_qkv(layer)
    for t in _triples(layer):
        st = pt_layer._stacked(t)
        assert st.data_ptr() == t[0].data_ptr() and tuple(st.shape) == (3,) + tuple(t[0].shape) and torch.equal(st, torch.stack(t))
    assert [id(p) for p in layer.parameters()] == params
    assert all(torch.equal(v, before[k]) for k, v in layer.state_dict().items())
    with torch.no_grad():
        layer.linear_k.weight.add_(1.0)                                         # an in-place update (an optimizer's) is seen through the view
    assert torch.equal(pt_layer._stacked(_triples(layer)[0])[1], layer.linear_k.weight)


def test_flat_state_keeps_the_projections_adjacent():
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(8, 8), blocks.PointTransformerLayer(128, 128, 8, 16), blocks.PointTransformerLayer(32, 32, 8, 16))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    state = distributed.FlatState([net], opt)
    assert sorted(id(p) for p in state.params) == sorted(id(p) for p in net.parameters())
    for t in _triples(net[1]):                                                  # sizes are multiples of FlatState.ALIGN: back to back
        assert pt_layer._stacked(t).data_ptr() == t[0].data_ptr()
    w, b = _triples(net[2])                                                     # 32 x 32 weights are, 32-float biases are not (padding between them): a copy, still right
    assert pt_layer._stacked(w).data_ptr() == w[0].data_ptr()
    assert torch.equal(pt_layer._stacked(b), torch.stack(b))
    assert all(torch.equal(v, before[k]) for k, v in net.state_dict().items())
