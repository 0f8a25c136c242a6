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


def test_adjoin_twice_keeps_the_storage():
    """a second adjoin_qkv (another GraphedTrainStep on the same model) must not move weights an earlier capture holds by address"""
    torch.manual_seed(2)
    layer = blocks.PointTransformerLayer(64, 64, 8, 16)
    pt_layer.adjoin_qkv(layer)
    where = [[t.data_ptr() for t in tri] for tri in _triples(layer)]
    pt_layer.adjoin_qkv(layer)
    assert [[t.data_ptr() for t in tri] for tri in _triples(layer)] == where


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


def test_state_dict_round_trip_keeps_the_layout_and_an_optimizer_step_is_seen_through_the_view():
    torch.manual_seed(2)
    a, b = blocks.PointTransformerLayer(128, 128, 8, 16), blocks.PointTransformerLayer(128, 128, 8, 16)
    pt_layer.adjoin_qkv(a)
    a.load_state_dict(b.state_dict())                                           # copies in place: the three weights stay back to back
    w = _triples(a)[0]
    st = pt_layer._stacked(w)
    assert st.data_ptr() == w[0].data_ptr() and torch.equal(st, torch.stack([b.linear_q.weight, b.linear_k.weight, b.linear_v.weight]))
    opt = torch.optim.SGD(a.parameters(), lr=0.5, momentum=0.9)
    for p in a.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    st2 = pt_layer._stacked(_triples(a)[0])
    assert st2.data_ptr() == a.linear_q.weight.data_ptr()
    assert torch.allclose(st2, torch.stack([b.linear_q.weight, b.linear_k.weight, b.linear_v.weight]) - 0.5)
    # a deep copy is a model of its own: its projections may or may not be adjacent, the stack is right either way
    import copy
    c = copy.deepcopy(a)
    assert torch.equal(pt_layer._stacked(_triples(c)[0]), torch.stack(_triples(c)[0]))
