"""CPU: hotpath.Pipeline.plan — every stream layout of the software-pipelined step covers every stage exactly once and never issues a stage before
what it needs (no device: the planning is pure)."""
import pytest

KPCONV = ["knnquery_k16", "queryandgroup", "kpconv_fwd", "cbl_knnquery_k36", "cbl_neighbor_transpose", "cbl_mining_loss_fwd", "cbl_mining_loss_bwd",
          "neighbor_transpose_k16", "queryandgroup_bwd", "kpconv_bwd"]
PT = ["knnquery_k16", "pt_layer_fwd", "cbl_knnquery_k36", "cbl_neighbor_transpose", "cbl_mining_loss_fwd", "cbl_mining_loss_bwd", "neighbor_transpose_k16",
      "pt_layer_bwd"]
FORWARD_ONLY = ["knnquery_k16", "queryandgroup", "kpconv_fwd", "cbl_knnquery_k36", "cbl_neighbor_transpose", "cbl_mining_loss_fwd", "cbl_mining_loss_bwd"]
LAYOUTS = ["tables", "split_t36_first", "alt_bwd", "pair_split", "pair_alt_bwd"]
# pair layouts: the K = 36 table stage builds the block's K = 16 table with it (hotpath.stages(pair_tables=True)), so the block's table stage — a registry hit —
# and its consumers must also sit behind the K = 36 table stage
PAIR_NEEDS = {"neighbor_transpose_k16": ["cbl_neighbor_transpose"]}
# stage -> stages that must have been issued on the same stream earlier, or on another stream behind an event the stage's segment waits for
NEEDS = {"queryandgroup": ["knnquery_k16"], "kpconv_fwd": ["knnquery_k16"], "pt_layer_fwd": ["knnquery_k16"], "cbl_neighbor_transpose": ["cbl_knnquery_k36"],
         "cbl_mining_loss_fwd": ["cbl_knnquery_k36"], "cbl_mining_loss_bwd": ["cbl_mining_loss_fwd", "cbl_neighbor_transpose"],
         "neighbor_transpose_k16": ["knnquery_k16"], "queryandgroup_bwd": ["queryandgroup", "neighbor_transpose_k16"], "kpconv_bwd": ["kpconv_fwd", "neighbor_transpose_k16"],
         "pt_layer_bwd": ["pt_layer_fwd", "neighbor_transpose_k16"]}


@pytest.mark.parametrize("names", [KPCONV, PT, FORWARD_ONLY])
@pytest.mark.parametrize("layout", LAYOUTS)
def test_layout_covers_every_stage_once_and_orders_them(names, layout):
    from contrastboundary_amd.hotpath import Pipeline
    streams, segs = Pipeline.plan(names, layout)
    assert streams[0] == "search" and len(streams) <= 4                          # a process has four hardware queues
    issued = [i for sg in segs for i in sg[2]]
    assert sorted(issued) == list(range(len(names)))                            # every stage exactly once
    recorded_by = {sg[4]: k for k, sg in enumerate(segs) if sg[4]}
    resolve = lambda sname: sname.replace("*", "0")
    assert all(resolve(sg[1]) in streams for sg in segs)
    for k, sg in enumerate(segs):
        assert all(w in recorded_by and recorded_by[w] < k for w in sg[3])      # an event is recorded before anything waits for it
        # what is visible to this segment: earlier segments of its stream, and (transitively) everything in front of the events it waits for
        visible, todo = set(), [k]
        seen = set()
        while todo:
            cur = todo.pop()
            if cur in seen:
                continue
            seen.add(cur)
            for j in range(cur):
                if resolve(segs[j][1]) == resolve(segs[cur][1]):
                    todo.append(j); visible.update(segs[j][2])
            for w in segs[cur][3]:
                j = recorded_by[w]
                todo.append(j); visible.update(segs[j][2])
        for pos, i in enumerate(sg[2]):
            have = visible | set(sg[2][:pos])
            for need in NEEDS.get(names[i], []) + (PAIR_NEEDS.get(names[i], []) if layout.startswith("pair_") else []):
                if need in names:
                    assert names.index(need) in have, (layout, names[i], "before", need)


def test_unknown_layout_is_refused():
    from contrastboundary_amd.hotpath import Pipeline
    with pytest.raises(ValueError):
        Pipeline.plan(KPCONV, "no such layout")
