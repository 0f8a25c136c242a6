"""`python bench.py --block pt`: the step object it times (KNN -> PointTransformerLayer forward + backward -> CBL head, hipGraph replay,
consecutive steps software-pipelined) checked directly:
    search      knnquery_cuda_kernel.cu:65-111   every query bit-exact against the oracle
    a4 layer    blocks.py:31-44                  the fused path (csrc/attention.hip) against the UNFUSED mirror of the same layer — the reference's own
                                                 sequence of ops (subtraction / Linear / BatchNorm / softmax / aggregation), which the goldens of
                                                 test_gpu_blocks.py pin to the reference at C = 32 ... 512 — on the bench scene itself: output 1e-4
                                                 elementwise, gradients 1e-3 in L2 (ReLU masks, see `close`)
    CBL head    heads.py:185-246                 loss and gradient against the oracle, 1e-4
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
N, C, K = 40960, 64, 16


def close(got, ref, what, tol=1e-4, gradient=False):
    """outputs: |got - ref| <= tol (|ref| + scale) for EVERY entry.
    gradients: relative L2 error <= 1e-3.  The layer has two ReLUs behind train-mode BatchNorms over n*K rows (42 M and 5 M activations here); an
    activation within rounding of zero takes a different mask in two fp32 evaluations that sum in different orders (the MFMA chain of the fused
    kernel, the lane tree of the streaming Linear, a library GEMM), and every such flip moves the gradient entries it feeds by O(1e-2) of their
    scale — measured here against the unfused layer: 126 of 2.6 M input-gradient entries beyond the elementwise bound, 5e-4 relative L2 error on
    the weight gradients (sums over all points, so a few flips touch every entry a little).  The reference's own fp32 and fp64 runs differ the
    same way (tests/golden/gen_model_goldens.py: 1.6e-3 on its first layer's weight gradient)."""
    got, ref = got.detach().double().cpu().numpy(), ref.detach().double().cpu().numpy()
    l2 = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))
    if gradient:
        assert l2 <= 1e-3, f"{what}: relative L2 error {l2:.2e}"
        return
    bound = tol * (np.abs(ref) + np.abs(ref).max())
    bad = np.abs(got - ref) > bound
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} entries beyond the bound (worst {float((np.abs(got - ref) / bound).max()):.1f}x), relative L2 error {l2:.2e}"


@pytest.mark.parametrize("pipeline", [True, False])
def test_the_pt_step_bench_times(pipeline):
    import bench
    from contrastboundary_amd import hotpath
    from oracle import cbl_oracle
    from tests import oracle_lib as O
    args = bench.parse(["--block", "pt"])
    scene = hotpath.Scene.synthetic(N, C, seed=0, b=1)
    step = bench.Step(scene, K, True, args, overlap=True, pipeline=pipeline)
    assert step.names[:2] == ["knnquery_k16", "pt_layer_fwd"] and step.names[-1] == "pt_layer_bwd"
    bench.settle(step, 0.1)
    step.capture()
    assert (step.pipe is not None) if pipeline else (step.graph is not None)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    sc = hotpath.Scene.synthetic_numpy(N, C, 0)
    xyz, off = sc["xyz"], sc["offset"]
    idx = np.zeros((N, K), np.int32); d2 = np.zeros((N, K), np.float32)
    O.lib().oracle_knnquery_omp(N, K, O.P(xyz), O.P(xyz), O.P(off), O.P(off), O.P(idx), O.P(d2), 0)
    widx = np.zeros((N, hotpath.CBL_NSAMPLE), np.int32); wd2 = np.zeros((N, hotpath.CBL_NSAMPLE), np.float32)
    O.lib().oracle_knnquery_omp(N, hotpath.CBL_NSAMPLE, O.P(xyz), O.P(xyz), O.P(off), O.P(off), O.P(widx), O.P(wd2), 0)
    rloss, rgrad, _ = cbl_oracle.point_contrast(sc["latent"], np.eye(13, dtype=np.float32)[sc["labels"]], widx, temperature=1.0, weight=0.1)
    # the unfused mirror of the same layer, same weights, train mode (running statistics do not enter the outputs)
    layer = hotpath.pt_layer(scene)
    up = scene.upstream(K)["grad_kpconv"]
    layer.fused = False
    try:
        x = scene.feat.detach().requires_grad_(True)
        ref_out = layer([scene.xyz, x, scene.offset], idx=torch.from_numpy(idx).cuda())
        ref_grads = torch.autograd.grad(ref_out, [x] + list(layer.parameters()), up)
    finally:
        layer.fused = True
    for s in step.states:
        assert np.array_equal(s["idx"].cpu().numpy(), idx) and np.array_equal(s["dist2"].cpu().numpy().view(np.uint32), d2.view(np.uint32))
        close(s["pt_out"], ref_out, "PointTransformerLayer output")
        close(s["grad_feat_pt"], ref_grads[0], "gradient w.r.t. the input features", gradient=True)
        named = dict(zip([nm for nm, _ in layer.named_parameters()], zip(s["grad_params_pt"], ref_grads[1:])))
        vscale = float(named["linear_v.bias"][1].abs().max())
        for name, (got, ref) in named.items():
            if name in ("linear_q.bias", "linear_k.bias", "linear_p.0.bias", "linear_w.2.bias", "linear_w.5.bias"):
                # biases in front of a train-mode BatchNorm (a constant added to x_q / x_k shifts every w[i,k,c] of a channel by the same amount;
                # Linear(3,3) and Linear(C,C/8) feed BatchNorm directly): the normalisation removes them, their gradients are exactly zero in
                # exact arithmetic and pure rounding noise in any fp32 run; the last Linear's bias shifts all K logits of a point alike,
                # which the softmax over K ignores: zero as well
                assert float(got.abs().max()) <= 1e-4 * vscale and float(ref.abs().max()) <= 1e-4 * vscale, name
                continue
            close(got, ref, f"gradient of {name}", gradient=True)
        assert abs(s["cbl_loss"].item() - rloss) < 1e-4 * max(1.0, abs(rloss))
        assert np.allclose(s["cbl_grad"].cpu().numpy(), rgrad, rtol=1e-4, atol=1e-4 * np.abs(rgrad).max())
