"""The configuration bench.py TIMES, checked against the oracles directly (not through another schedule of ours):
S-room scene of 40960 points, seed 0, C = 64, K = 16 derived from the CBL head's K = 36 search, the CBL branch on a side stream, the
whole step captured in a hipGraph and replayed — `bench.Step`, the object bench.py's timed region calls.
    search        knnquery_cuda_kernel.cu:65-111        idx / dist2 of ALL 40960 queries bit-exact
    gather        pointops.py:79-100                    grouped (N,K,3+C) bit-exact
    KPConv        local_aggregation_operators.py:681-728   1e-4 (MFMA fma chain vs numpy's summation order)
    CBL head      heads.py:185-246                      loss 1e-4, gradient 1e-4 of its scale
    backward legs grouping_cuda_kernel.cu:16-25 (bit-exact: the gather sums in the reference loop's order), KPConv gradients 1e-4
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, C, K = 40960, 64, 16


@pytest.fixture(scope="module")
def oracle():
    from contrastboundary_amd import hotpath
    from oracle import cbl_oracle, local_aggregation_oracle as LA
    from tests import oracle_lib as O
    sc = hotpath.Scene.synthetic_numpy(N, C, 0)
    up = hotpath.Scene.upstream_numpy(N, C, K, 0)
    xyz, feat, off = sc["xyz"], sc["feat"], sc["offset"]
    lib, P = O.lib(), O.P

    def knn(k):
        idx = np.zeros((N, k), np.int32); d2 = np.zeros((N, k), np.float32)
        lib.oracle_knnquery_omp(N, k, P(xyz), P(xyz), P(off), P(off), P(idx), P(d2), 0)      # every query, all host cores
        return idx, d2
    r = {"sc": sc, "up": up}
    r["idx"], r["d2"] = knn(K)
    r["widx"], r["wd2"] = knn(hotpath.CBL_NSAMPLE)
    r["grouped"] = np.concatenate([xyz[r["idx"]] - xyz[:, None, :], feat[r["idx"]]], -1)
    r["kpconv"] = LA.kpconv(xyz, xyz, r["idx"], feat, sc["kernel_points"], sc["kernel_weights"], 0.12)
    r["loss"], r["grad"], _ = cbl_oracle.point_contrast(sc["latent"], np.eye(13, dtype=np.float32)[sc["labels"]], r["widx"], temperature=1.0, weight=0.1)
    r["g_group"] = O.grouping_backward(np.ascontiguousarray(up["grad_grouped"][..., 3:]), r["idx"], N)
    r["g_feat_kp"], r["g_kw"] = LA.kpconv_grads(xyz, xyz, r["idx"], feat, sc["kernel_points"], sc["kernel_weights"], 0.12, up["grad_kpconv"])
    return r


def check_state(s, o, backward):
    from contrastboundary_amd import hotpath
    cpu = lambda t: t.detach().cpu().numpy()
    assert np.array_equal(cpu(s["idx"]), o["idx"]), "K=16 neighbour indices differ from the oracle"
    assert np.array_equal(cpu(s["dist2"]).view(np.uint32), o["d2"].view(np.uint32)), "K=16 squared distances differ"
    # the CBL head's wide search is the set-exact variant: same neighbour SET and the same ascending distances per row
    widx = cpu(s["cbl_idx"])
    assert widx.shape == (N, hotpath.CBL_NSAMPLE)
    assert np.array_equal(np.sort(widx, 1), np.sort(o["widx"], 1)), "K=36 neighbour sets differ from the oracle"
    assert np.array_equal(cpu(s["grouped"]), o["grouped"]), "grouped tensor differs"
    kp = cpu(s["kpconv"])
    assert np.allclose(kp, o["kpconv"], rtol=1e-4, atol=1e-4 * np.abs(o["kpconv"]).max()), "KPConv output beyond 1e-4"
    assert abs(s["cbl_loss"].item() - o["loss"]) < 1e-4 * max(1.0, abs(o["loss"])), "CBL loss beyond 1e-4"
    g = cpu(s["cbl_grad"])
    assert np.allclose(g, o["grad"], rtol=1e-4, atol=1e-4 * np.abs(o["grad"]).max()), "CBL gradient beyond 1e-4"
    if backward:
        assert np.array_equal(cpu(s["grad_feat_group"]), o["g_group"]), "grouping backward (K4) differs from the reference loop's sums"
        gf = cpu(s["grad_feat_kpconv"])
        assert np.allclose(gf, o["g_feat_kp"], rtol=1e-4, atol=1e-4 * np.abs(o["g_feat_kp"]).max()), "KPConv feature gradient beyond 1e-4"
        gk = cpu(s["grad_kernel_weights"])
        assert np.allclose(gk, o["g_kw"], rtol=1e-4, atol=1e-4 * np.abs(o["g_kw"]).max()), "KPConv kernel-weight gradient beyond 1e-4"


@pytest.mark.parametrize("backward,pipeline", [(True, True), (False, True), (True, False), (False, False)])
def test_the_step_bench_times_against_the_oracles(oracle, backward, pipeline):
    """pipeline = True: what `python bench.py` runs by default — consecutive steps software-pipelined over four streams (the search of step i+1
    beside the forward of step i and the backward of step i-1), rotating through the pipeline's output slots: EVERY slot is checked after a
    number of steps that is not a multiple of the slot count (a step reading a neighbour table, an order or a transposed table that a later
    search has already overwritten would show up here)."""
    import bench
    from contrastboundary_amd import hotpath
    args = bench.parse([])
    scene = hotpath.Scene.synthetic(N, C, seed=0, b=1)
    step = bench.Step(scene, K, backward, args, overlap=True, pipeline=pipeline)
    bench.settle(step, 0.1)
    step.capture()                                                    # the hipGraph(s) bench.py replays; a failed capture fails the test
    assert (step.pipe is not None) if pipeline else (step.graph is not None)
    for _ in range(7):
        step()                                                        # what the timed region calls
    torch.cuda.synchronize()
    assert len(step.states) == (step.pipe.SLOTS if pipeline else 1)
    for st in step.states:
        check_state(st, oracle, backward)
    if pipeline:
        return
    # the in-order step with events inside its graph (where the per-stage times come from) computes the same thing
    st_in, ms, how = bench.stage_times(scene, K, backward, args, reps=2)
    assert len(ms) == len(st_in.stages) and all(np.isfinite(ms)) and min(ms) >= 0
    check_state(st_in.state, oracle, backward)


@pytest.mark.parametrize("layout,slots", [("tables", 2), ("split_t36_first", 4), ("alt_bwd", 4), ("pair_split", 3), ("pair_alt_bwd", 4)])
def test_every_pipeline_layout_computes_the_step(oracle, layout, slots):
    """hotpath.Pipeline's stream layouts differ in what runs beside what, never in what is computed: every slot of every layout against the oracles
    (the default, "split_t36_first" with three slots, is also the pipeline case of the test above)"""
    import bench
    from contrastboundary_amd import hotpath
    args = bench.parse([])
    scene = hotpath.Scene.synthetic(N, C, seed=0, b=1)
    # "pair_" layouts: the K = 36 table stage builds the block's K = 16 table with it (cbl_neighbor_transpose_pair), the backward chain waits for that stage
    step = bench.Step(scene, K, True, args, overlap=True, pipeline=False, pair_tables=layout.startswith("pair_"))
    bench.settle(step, 0.1)
    with pytest.raises(ValueError):                                  # a layout that does not order the backward behind the pair build is refused, not raced
        hotpath.Pipeline(step.sched, layout="split_t36_first" if layout.startswith("pair_") else "pair_split", slots=slots)
    pipe = hotpath.Pipeline(step.sched, layout=layout, slots=slots)
    pipe.capture()
    assert pipe.layout == layout and pipe.SLOTS == slots and len(pipe.states) == slots
    for _ in range(2 * slots + 1):
        pipe.step()
    pipe.join()
    torch.cuda.synchronize()
    for st in pipe.states:
        check_state(st, oracle, True)
    with pytest.raises(ValueError):
        hotpath.Pipeline(step.sched, layout="no such layout")
