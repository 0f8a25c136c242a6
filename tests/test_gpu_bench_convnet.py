"""The object `bench.py --workload convnet` times (contrastboundary_amd/convnet_path.py), checked against the oracles:
    pyramid       neighbors.cpp:213-336 + grid_subsampling.cpp:114 via oracle/tfops_oracle.c (pinned to oracle/_ref)      bit-exact
    AdaptiveWeight local_aggregation_operators.py:360-484 via oracle/local_aggregation_oracle.py (parity unpinned: TF absent)  1e-4
    scene labels  heads/head.py:25-49 via oracle/cbl_oracle.py                                                            exact
    TF CBL        heads/head.py:462-807 via oracle/cbl_oracle.py (parity unpinned)                                          loss 1e-4, gradient 1e-4 of its scale
at 30000 points (the oracles finish in seconds); at BASELINE's 200000 points through size-independent properties.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_convnet_step_against_the_oracles():
    from contrastboundary_amd import convnet_path as CP
    from oracle import cbl_oracle as C, local_aggregation_oracle as LA
    from tests import oracle_lib as O
    n = 30000
    scene = CP.ConvNetScene(n, seed=2)
    state = {}
    for _ in range(2):                                               # the second run re-uses workspaces and per-layer arrays, like the timed steps
        CP.run_once(scene, state)
    torch.cuda.synchronize()
    cpu = lambda t: t.detach().cpu().numpy()
    a = CP.ConvNetScene.synthetic_numpy(n, 2)
    p, l, r, dl = a["points"], a["lengths"], CP.DL0 * CP.DENSITY / 2.0, CP.DL0
    pyr = state["pyr"]
    labels = [a["labels"]]
    for lay in range(CP.NUM_LAYERS):
        np.testing.assert_array_equal(cpu(pyr["points"][lay]).view(np.uint32), p.view(np.uint32))
        ref, _, mc = O.radius_neighbors(p, p, l, l, r, CP.LIMITS[lay])
        nb = ref[:, :min(mc, CP.LIMITS[lay])]
        np.testing.assert_array_equal(cpu(pyr["neighbors"][lay]), nb)
        # AdaptiveWeight forward + backward on this layer
        arr = CP.ConvNetScene.layer_arrays_numpy(a["seeds"][lay], len(p), CP.WIDTHS[lay])
        W, b = a["fc_weight"][lay], a["fc_bias"][lay]
        out = LA.adaptive_weight(p, p, nb, arr["feat"], r, W, b, "mean")
        got = cpu(state["aw_out%d" % lay])
        np.testing.assert_allclose(got, out, rtol=1e-4, atol=1e-4 * np.abs(out).max())
        gf, gw, gb = LA.adaptive_weight_grads(p, p, nb, arr["feat"], r, W, b, arr["grad"], "mean")
        hf, hw, hb = [cpu(t) for t in state["aw_grads%d" % lay]]
        np.testing.assert_allclose(hf, gf, rtol=1e-4, atol=1e-4 * np.abs(gf).max())
        # parameter gradients are sums over every (point, neighbour) pair: 1e-4 of their scale
        np.testing.assert_allclose(hw, gw, rtol=1e-4, atol=1e-4 * np.abs(gw).max())
        np.testing.assert_allclose(hb, gb, rtol=1e-4, atol=1e-4 * np.abs(gb).max())
        # scene labels + CBL
        np.testing.assert_array_equal(cpu(state["labels"][lay]), labels[lay])
        rl, rg, rm = C.tf_contrast(arr["latent"], labels[lay], nb, temperature=1.0, weight=0.1)
        assert abs(state["cbl_loss%d" % lay].item() - rl) < 1e-4 * max(1.0, abs(rl))
        np.testing.assert_array_equal(cpu(state["cbl_mask%d" % lay]) > 0, rm > 0)
        np.testing.assert_allclose(cpu(state["cbl_grad%d" % lay]), rg, rtol=1e-4, atol=1e-4 * max(np.abs(rg).max(), 1e-12))
        if lay == CP.NUM_LAYERS - 1:
            break
        pp, pl = O.grid_subsampling(p, l, 2 * dl)
        refp, _, mcp = O.radius_neighbors(pp, p, pl, l, r, CP.LIMITS[lay])
        pools = refp[:, :min(mcp, CP.LIMITS[lay])]
        np.testing.assert_array_equal(cpu(pyr["pools"][lay]), pools)
        labels.append(C.tf_scene_label(labels[lay], pools, CP.NUM_CLASSES, "max"))
        p, l, r, dl = pp, pl, 2 * r, 2 * dl


@pytest.mark.parametrize("n", [30000, 200000])
def test_native_layers_call_equals_the_ops_issued_one_by_one(n):
    """cbl_convnet_step (convnet_path.NativeLayers: every layer's AdaptiveWeight forward + backward, scene labels and contrast head as ONE native call) against
    the same scene through the autograd Functions (which the test above holds to the oracles): the same kernels in the same order — bit-identical wherever
    the op-by-op path takes the transposed table too (tables of >= 65536 pairs; below that it scatters with float atomics: 1e-5 there), run twice (buffers
    and workspace are re-used), and a second scene of other sizes through the same runner"""
    from contrastboundary_amd import convnet_path as CP, pointops
    scene = CP.ConvNetScene(n, seed=2)
    state = CP.run_once(scene)
    runner = CP.NativeLayers(scene)
    for rep in range(2):
        out = runner(state["pyr"])
    torch.cuda.synchronize()
    for lay in range(CP.NUM_LAYERS):
        exact = state["pyr"]["neighbors"][lay].numel() >= pointops.TRANSPOSE_MIN_PAIRS
        same = (lambda a, b: torch.equal(a, b)) if exact else (lambda a, b: torch.allclose(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max())))
        assert torch.equal(out["aw_out"][lay], state["aw_out%d" % lay]), lay
        for got, ref in zip(out["aw_grads"][lay], state["aw_grads%d" % lay]):
            assert got.shape == ref.shape and same(got, ref), (lay, exact)
        assert torch.equal(out["labels"][lay].long(), state["labels"][lay].long()), lay
        assert torch.equal(out["cbl_loss"][lay].reshape(()), state["cbl_loss%d" % lay].reshape(())), lay
        assert torch.equal(out["cbl_mask"][lay], state["cbl_mask%d" % lay]), lay
        assert same(out["cbl_grad"][lay], state["cbl_grad%d" % lay]), (lay, exact)
    # the native stage list (pyramid + one call) is what bench.py times
    st = CP.run_once(scene, stage_list=CP.native_stages(scene, runner=runner))
    torch.cuda.synchronize()
    assert torch.equal(st["native"]["aw_out"][0], state["aw_out0"])
    if n == 30000:
        other = CP.ConvNetScene(21000, seed=5)
        st2 = CP.run_once(other)
        out2 = CP.NativeLayers(other)(st2["pyr"])
        torch.cuda.synchronize()
        assert torch.equal(out2["aw_out"][1], st2["aw_out1"]) and torch.equal(out2["cbl_mask"][0], st2["cbl_mask0"])


def test_convnet_step_full_size_properties():
    """N = 200000 (BASELINE config C5): linearity of AdaptiveWeight in its features, rows sorted and inside the ball, idempotent stage labels"""
    from contrastboundary_amd import convnet_path as CP, local_aggregation as LA
    scene = CP.ConvNetScene(200000, seed=0)
    state = CP.run_once(scene)
    pyr = state["pyr"]
    assert [int(p.shape[0]) for p in pyr["points"]][0] == 200000 and all(pyr["points"][i + 1].shape[0] < pyr["points"][i].shape[0] for i in range(4))
    for lay in range(CP.NUM_LAYERS):
        q, nb = pyr["points"][lay], pyr["neighbors"][lay]
        nrows = q.shape[0]
        assert (nb[:, 0] == torch.arange(nrows, device="cuda")).all()          # self first
        arr = scene.layer_arrays(lay, nrows)
        r = CP.DL0 * CP.DENSITY / 2.0 * 2 ** lay
        f = arr["feat"]
        o1 = LA.adaptive_weight(q, q, nb, f, r, scene.fc_weight[lay], scene.fc_bias[lay], "mean")
        o2 = LA.adaptive_weight(q, q, nb, 2.0 * f, r, scene.fc_weight[lay], scene.fc_bias[lay], "mean")
        assert torch.equal(o2, 2.0 * o1)                                        # linear in the features: a factor 2 is exact in fp32
        assert torch.equal(o1, state["aw_out%d" % lay])                         # deterministic forward
        assert torch.isfinite(state["cbl_loss%d" % lay]).all() and torch.isfinite(state["cbl_grad%d" % lay]).all()
        # points without a loss term (mask 0) carry no centre gradient of their own unless they are someone's neighbour: total gradient sums to ~0
        g = state["cbl_grad%d" % lay]
        assert abs(float(g.sum())) <= 1e-3 * float(g.abs().sum()) + 1e-6      # every pair pushes centre and neighbour by opposite amounts


def test_bench_convnet_json_contract():
    e = dict(os.environ)
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(key, None)
    r = subprocess.run([sys.executable, "bench.py", "--workload", "convnet", "--points", "60000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out
    assert out["n_gpus"] == 1 and out["steps"] == 3 and "ConvNet" in out["config"]["workload"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    assert rf["adaptive_weight"]["launch_us"] > 0 and "pyramid_radius_grid" in rf["stage_ms"]
