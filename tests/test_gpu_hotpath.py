"""The bench step's two-stream schedule (hotpath.Schedule) must produce what the in-order schedule produces."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_overlapped_schedule_equals_in_order_schedule():
    from contrastboundary_amd import hotpath
    sc = hotpath.Scene.synthetic(8192, 32, seed=3)
    st = hotpath.stages(sc, 16)
    ref = hotpath.run_once(sc, 16, {})
    torch.cuda.synchronize()
    for side_after in (None, "knnquery_k16"):
        sched = hotpath.Schedule(st, overlap=True)
        state = {}
        for _ in range(3):                                   # repeated steps reuse and free side-stream buffers
            sched.run(state, side_after=side_after)
        torch.cuda.synchronize()
        assert torch.equal(state["idx"], ref["idx"]) and torch.equal(state["cbl_idx"], ref["cbl_idx"])
        assert torch.equal(state["grouped"], ref["grouped"]) and torch.equal(state["kpconv"], ref["kpconv"])
        assert abs(float(state["cbl_loss"].detach()) - float(ref["cbl_loss"].detach())) <= 1e-6 * abs(float(ref["cbl_loss"].detach()))
        torch.testing.assert_close(state["cbl_grad"], ref["cbl_grad"], rtol=1e-4, atol=1e-7)   # atomics: summation order differs


def test_schedule_records_stage_events_on_the_stage_stream():
    from contrastboundary_amd import hotpath
    sc = hotpath.Scene.synthetic(4096, 32, seed=1)
    st = hotpath.stages(sc, 16)
    sched = hotpath.Schedule(st, overlap=True)
    state = {}
    sched.run(state)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in st]
    sched.run(state, ev)
    torch.cuda.synchronize()
    assert all(a.elapsed_time(b) > 0.0 for a, b in ev)


@pytest.mark.parametrize("overlap", [False, True])
def test_step_replays_from_a_hipgraph(overlap):
    """bench.py captures the step (nested searches, the CBL branch on a side stream, autograd backward of the CBL loss included) once
    and replays it"""
    from contrastboundary_amd import hotpath
    sc = hotpath.Scene.synthetic(16384, 32, seed=7)
    st = hotpath.stages(sc, 16)
    ref = hotpath.run_once(sc, 16, {})
    sched = hotpath.Schedule(st, overlap=overlap, hints=hotpath.search_hints(sc))
    gstate = {}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            sched.run(gstate)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sched.run(gstate)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(gstate["idx"], ref["idx"]) and torch.equal(gstate["grouped"], ref["grouped"]) and torch.equal(gstate["kpconv"], ref["kpconv"])
    assert abs(float(gstate["cbl_loss"].detach()) - float(ref["cbl_loss"].detach())) <= 1e-5 * abs(float(ref["cbl_loss"].detach()))
    torch.testing.assert_close(gstate["cbl_grad"], ref["cbl_grad"], rtol=1e-4, atol=1e-7)
