"""GPU parity of the full network + criterion (C4: Point Transformer + CBL) against the reference's own model run on CPU
(tests/golden/model_pytorch.npz, made by gen_model_goldens.py: reference code + CPU oracle KNN / FPS).  The mirror is built under
the same seed, which reproduces the reference's 7.8 M initial parameters (checked by checksum).  Indices (FPS, KNN) are bit-exact,
so the two runs differ only by float summation order in the dense layers.  Against the reference's fp32 run: logits / losses within 2e-3
relative, parameter gradients within 2 % in L2 (ReLU / max-pool ties can flip single entries); against the reference run in FLOAT64 (`*64`
goldens): every logit and loss term within 1e-4."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_pytorch.npz"))
CASES = sorted({k.split("/")[0] for k in G.files})


def shipped_config(M):
    return M.Config({"base_fdim": 32, "nsample": [36, 24, 24, 24, 24], "nstride": [4, 4, 4, 4], "ignore_label": 255, "voxel_size": 0.04,
                     "contrast": {"stage": "Ua", "contrast": "softnn", "ftype": "latent", "sample": "label", "pos": "cnt", "dist": "l2",
                                  "temperature": 1, "weight": "w.1"},
                     "multi": {"stage": "Ua", "ftype": "latent", "combine": "concat"}})


def build(case):
    from contrastboundary_amd import pointtransformer_seg as M
    g = lambda f: G[f"{case}/{f}"]
    torch.manual_seed(int(g("seed")))
    cfg = shipped_config(M)
    model = M.pointtransformer_seg_repro(c=6, k=13, config=cfg)
    crit = M.Loss(cfg)
    return M, model, crit, g


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


@pytest.mark.parametrize("case", CASES)
def test_same_seed_gives_the_reference_parameters(case):
    M, model, crit, g = build(case)
    s = sum(float(v.double().abs().sum()) for k, v in model.state_dict().items() if v.dtype.is_floating_point and "running" not in k)
    assert abs(s - float(g("param_abs_sum"))) < 1e-9 * float(g("param_abs_sum"))
    assert sum(p.numel() for p in model.parameters()) == 7800497
    # same names and shapes: a reference checkpoint loads into the mirror (and back)
    assert [f"{k}:{tuple(v.shape)}" for k, v in model.state_dict().items()] == list(g("state_dict_keys"))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_network_and_criterion_match_reference(case):
    M, model, crit, g = build(case)
    model = model.cuda().train()
    inputs = {"points": torch.from_numpy(g("xyz")).cuda(), "features": torch.from_numpy(g("feat")).cuda(), "offset": torch.from_numpy(g("offset")).cuda()}
    target = torch.from_numpy(g("target")).cuda()
    logits, stage_list, loss, nc = M.forward_and_loss(model, crit, inputs, target)
    loss.sum().backward()
    np.testing.assert_array_equal([st["p_out"].shape[0] for st in stage_list["up"]], g("stage_sizes"))
    assert rel_l2(logits.detach().cpu().numpy(), g("logits")) < 2e-3
    np.testing.assert_allclose(loss.detach().cpu().numpy(), g("loss"), rtol=2e-3, atol=1e-5)
    assert rel_l2(model.enc1[0].linear.weight.grad.cpu().numpy(), g("grad_first")) < 2e-2
    assert rel_l2(model.head.cls.weight.grad.cpu().numpy(), g("grad_last")) < 2e-2
    # against the reference network run in float64 on the same inputs and indices (`*64`): north_star's 1e-4 on what the network outputs.
    # Every logit within 1e-4 of the logits' scale, every loss term within 1e-4 relative.  The parameter gradients pass through ~60 ReLU /
    # max-pool / BatchNorm layers: an activation within rounding of zero flips its mask between two fp32 runs (the reference's own fp32 run
    # differs from its fp64 run the same way: 1.6e-3 in L2 on the first layer's weight gradient, 3e-6 on the classifier's), so the first
    # layer's gradient (measured 0.85e-2: a handful of flipped masks among 60 layers, amplified by the BatchNorm backward) is held to 2e-2 in L2
    # as against the fp32 run, and the last layer's to 1e-4.
    lg = logits.detach().cpu().numpy().astype(np.float64)
    assert np.abs(lg - g("logits64")).max() <= 1e-4 * np.abs(g("logits64")).max(), float(np.abs(lg - g("logits64")).max() / np.abs(g("logits64")).max())
    np.testing.assert_allclose(loss.detach().cpu().numpy(), g("loss64"), rtol=1e-4, atol=1e-6)
    assert rel_l2(model.enc1[0].linear.weight.grad.cpu().numpy(), g("grad_first64")) < 2e-2
    assert rel_l2(model.head.cls.weight.grad.cpu().numpy(), g("grad_last64")) < 1e-4
    # neighbour cache: the reference issued 57 knnquery launches for this step; the mirror's blocks ask 39 times (one search per
    # layer instead of two) and 26 of those are distinct: 5 self + 4 down + 4 up (k=3) + 4 multi-head (k=1) + 5 CBL + 4 sub-scene
    assert int(g("ref_knn_calls")) == 57
    assert (nc.misses, nc.hits) == (26, 13), (nc.hits, nc.misses)


@pytest.mark.gpu
def test_neighbor_cache_does_not_change_the_forward():
    M, model, crit, g = build(CASES[0])
    model = model.cuda().train()
    inputs = {"points": torch.from_numpy(g("xyz")).cuda(), "features": torch.from_numpy(g("feat")).cuda(), "offset": torch.from_numpy(g("offset")).cuda()}
    target = torch.from_numpy(g("target")).cuda()
    with torch.no_grad():
        a, _, la, _ = M.forward_and_loss(model, crit, inputs, target)
        b, sl = model(inputs)
        lb = crit(b, target, sl)
        c, sl2 = model(inputs)
        lc = crit(c, target, sl2)
    # the same pass twice: bit-identical (no atomics anywhere in the forward)
    assert torch.equal(b, c) and torch.equal(lb, lc)
    # with the cache the widest search of a geometry runs first and its cell order becomes the processing order; without it the layer's own (narrower)
    # search does — another permutation, hence another summation order of the BatchNorm statistics in csrc/pt_layer.hip: equal up to fp32 rounding
    # (same indices, same values; measured 5e-7 on the first layer, 1e-5 on the logits)
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 1e-4 * scale and torch.allclose(la, lb, rtol=1e-4, atol=1e-6)
    # ... and that IS the only difference: with the same processing order in both runs (index order) the cached and the uncached forward are bit-identical
    from contrastboundary_amd import neighbor_state
    with neighbor_state.natural_order(), torch.no_grad():
        a, _, la, _ = M.forward_and_loss(model, crit, inputs, target)
        b, sl = model(inputs)
        lb = crit(b, target, sl)
    assert torch.equal(a, b) and torch.equal(la, lb)


@pytest.mark.gpu
def test_geometry_prefetch_on_a_side_stream_answers_every_request():
    """all FPS + neighbour searches issued ahead on a side stream: the forward finds 39 + 4 requests answered, same results"""
    M, model, crit, g = build(CASES[1])
    model = model.cuda().train()
    inputs = {"points": torch.from_numpy(g("xyz")).cuda(), "features": torch.from_numpy(g("feat")).cuda(), "offset": torch.from_numpy(g("offset")).cuda()}
    target = torch.from_numpy(g("target")).cuda()
    with torch.no_grad():
        a, _, la, nc0 = M.forward_and_loss(model, crit, inputs, target)
        geom = M.prefetch_geometry(model, inputs, crit)
        b, _, lb, nc1 = M.forward_and_loss(model, crit, inputs, target, geometry=geom)
        c, _, lc, _ = M.forward_and_loss(model, crit, inputs, target, geometry=M.prefetch_geometry(model, inputs, crit))
    torch.cuda.synchronize()
    # prefetched twice: bit-identical; prefetched vs searched inside the forward: the searches of a geometry run in another sequence, the first one's
    # cell order becomes the processing order of csrc/pt_layer.hip's passes -> equal up to fp32 rounding (same indices everywhere: nc1 below)
    assert torch.equal(b, c) and torch.equal(lb, lc)
    assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) and torch.allclose(la, lb, rtol=1e-4, atol=1e-6)
    assert nc1.misses == 0 and nc1.hits == nc0.hits + nc0.misses
    # the same processing order (index order) on both sides: prefetched and inline geometry give the same bits
    from contrastboundary_amd import neighbor_state
    with neighbor_state.natural_order(), torch.no_grad():
        a, _, la, _ = M.forward_and_loss(model, crit, inputs, target)
        b, _, lb, _ = M.forward_and_loss(model, crit, inputs, target, geometry=M.prefetch_geometry(model, inputs, crit))
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(la, lb)


@pytest.mark.gpu
@pytest.mark.parametrize("natural", [False, True])
def test_graphed_training_step_matches_eager_steps(natural):
    """the whole step (forward, criterion, backward, SGD) replayed from a hipGraph with double-buffered static geometry gives the same
    training trajectory as eagerly issued steps (atomics make the two differ in the last bits only).  natural = True: both runs walk the points in
    index order (neighbor_state.natural_order), which removes the one legitimate difference between them that is NOT an atomic — the processing order
    of the fused attention layers' BatchNorm sums — so the comparison is held to the spread of two reruns of the same eager trajectory."""
    import contextlib
    import copy
    from contrastboundary_amd import neighbor_state
    with (neighbor_state.natural_order() if natural else contextlib.nullcontext()):
        _graphed_vs_eager(natural)


def _graphed_vs_eager(natural):
    import copy
    M, model, crit, g = build(CASES[0])
    model = model.cuda().train()
    twin = copy.deepcopy(model)
    inputs = {"points": torch.from_numpy(g("xyz")).cuda(), "features": torch.from_numpy(g("feat")).cuda(), "offset": torch.from_numpy(g("offset")).cuda()}
    target = torch.from_numpy(g("target")).cuda()
    # a second batch with the same shapes: the scene mirrored in x, labels rolled
    inputs2 = {"points": (inputs["points"] * torch.tensor([-1.0, 1.0, 1.0], device="cuda")).contiguous(), "features": inputs["features"].flip(0).contiguous(),
               "offset": inputs["offset"].clone()}
    target2 = target.roll(17)
    batches = [(inputs, target), (inputs2, target2), (inputs, target), (inputs2, target2)]

    opt_e = torch.optim.SGD(twin.parameters(), lr=0.002, momentum=0.9)
    eager = []
    for b_in, b_tg in batches:
        opt_e.zero_grad(set_to_none=True)
        _, _, loss, _ = M.forward_and_loss(twin, crit, b_in, b_tg)
        loss.sum().backward()
        opt_e.step()
        eager.append(loss.detach().cpu().numpy())

    opt_g = torch.optim.SGD(model.parameters(), lr=0.002, momentum=0.9)
    snapshot = copy.deepcopy(model.state_dict())
    step = M.GraphedTrainStep(model, crit, opt_g, inputs, target, warmup=1)      # its warm-up / capture steps train the model: rewind
    model.load_state_dict(snapshot)
    for st in opt_g.state.values():
        if "momentum_buffer" in st and st["momentum_buffer"] is not None:
            st["momentum_buffer"].zero_()
    graphed = []
    for b in batches[:step.depth]:                                                  # `depth` batches staged ahead of the one that runs
        step.stage(*b)
    for i in range(len(batches)):
        loss, _ = step.run()
        if i + step.depth < len(batches):
            step.stage(*batches[i + step.depth])
        graphed.append(loss.detach().cpu().numpy().copy())
    torch.cuda.synchronize()
    np.testing.assert_allclose(graphed[0], eager[0], rtol=1e-5 if natural else 2e-3, atol=1e-6 if natural else 1e-5)   # same parameters, same batch: the forward pass itself
    # Later steps: the two runs differ by ROUNDING at step 0 (the static geometry's searches and the per-forward cache's searches build different grids, so
    # the fused attention layers sum their BatchNorm statistics in a different processing order; fp32 atomics in the small deep stages), and training
    # amplifies a difference ~30-50 x per step on this scene (tools/traj_determinism.py: reruns of the SAME eager trajectory spread 8e-7 / 4e-5 / 1.4e-3
    # at steps 1 / 2 / 3).  A broken replay (stale buffers, a lost dependency) shows as O(1) garbage at step 1, far outside these bounds.
    # In index order on both sides only the atomics are left: 25 - 100 x the rerun spread above, and the weight drift bound of round 3.
    for (a, b), rtol in zip(zip(graphed[1:], eager[1:]), (1e-4, 2e-3, 3e-2) if natural else (5e-3, 3e-2, 2e-1)):
        np.testing.assert_allclose(a, b, rtol=rtol, atol=1e-4)
    w_g, w_e = model.enc1[0].linear.weight.detach(), twin.enc1[0].linear.weight.detach()
    assert float((w_g - w_e).norm() / w_e.norm()) < (1e-2 if natural else 2e-2)
    # the in-place refresh really holds the staged batch's geometry: bitwise the eagerly computed one
    from contrastboundary_amd import geometry
    step.stage(inputs2, target2)
    torch.cuda.synchronize()
    static = step.sets[(step.stage_turn - 1) % len(step.sets)]["geom"]
    fresh = M.prefetch_geometry(model, {"points": static.points, "offset": static.offset}, crit)
    torch.cuda.synchronize()
    assert torch.equal(static.points, inputs2["points"])
    for (a_outs, *_), (b_outs, *_) in zip(static.cache.store.values(), fresh.store.values()):
        for a, b in zip(a_outs, b_outs):
            assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,ignored", [(40960, 13, 0.0), (163840, 13, 0.1), (1000, 2, 0.5), (777, 64, 0.0), (5000, 20, 1.0)])
def test_cross_entropy_matches_torch_in_float64(n, k, ignored):
    """pointtransformer_seg.cross_entropy (cbl_cross_entropy_*: the criterion's nn.CrossEntropyLoss(ignore_index), pointtransformer_seg.py:20-22) against
    F.cross_entropy in float64: the loss, the gradient of the logits under an upstream factor, ignored points, the all-ignored batch (nan, like the library)"""
    from contrastboundary_amd import pointtransformer_seg as M
    torch.manual_seed(n + k)
    z = (torch.randn(n, k, device="cuda") * 3).requires_grad_(True)
    t = torch.randint(0, k, (n,), device="cuda")
    if ignored > 0:
        t[torch.rand(n, device="cuda") < ignored] = 255
    loss = M.cross_entropy(z, t, 255)
    (loss * 1.7).backward()
    z64 = z.detach().double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(z64, t, ignore_index=255)
    if ignored >= 1.0:
        assert torch.isnan(loss) and torch.isnan(ref)
        return
    (ref * 1.7).backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
    assert float((z.grad.double() - z64.grad).abs().max()) <= 1e-6 * float(z64.grad.abs().max())
    assert torch.equal(z.grad[t == 255], torch.zeros_like(z.grad[t == 255]))
    # twice the same bits (fixed-order fp64 combination of the partial sums)
    assert torch.equal(M.cross_entropy(z.detach(), t, 255), loss.detach())
