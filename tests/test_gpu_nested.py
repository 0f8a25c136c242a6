"""Narrower neighbour searches derived from a wider one over the same geometry (cbl_knnquery_prefix, neighbor_cache hints)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _rows_as_sets_equal(a, b, d2a, d2b):
    return torch.equal(d2a, d2b) and torch.equal(torch.sort(a, 1)[0], torch.sort(b, 1)[0])


@pytest.mark.parametrize("kind", ["room", "lattice", "duplicates"])
@pytest.mark.parametrize("ks,kw", [(16, 36), (8, 16), (5, 64)])
def test_prefix_equals_the_direct_search(kind, ks, kw):
    from contrastboundary_amd import pointops, synthetic as S
    rng = np.random.default_rng(ks * 100 + kw)
    if kind == "room":
        xyz = S.s_room(12000, seed=1)[0]
    elif kind == "lattice":                                                    # every distance tied: everything is replayed
        g = np.stack(np.meshgrid(np.arange(14), np.arange(14), np.arange(14), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * 0.1
        xyz = g[rng.permutation(len(g))]
    else:                                                                      # coincident points: zero distances, ties inside the lists
        base = rng.uniform(size=(3000, 3)).astype(np.float32)
        xyz = np.concatenate([base, base[:1500]])[rng.permutation(4500)]
    n = len(xyz)
    p = dev(xyz); off = dev(np.int32([n // 3, n]))
    wide_i, wide_d = pointops.knnquery_raw(kw, p, p, off, off, algo="set")
    for algo in ("auto", "set"):
        want_i, want_d = pointops.knnquery_raw(ks, p, p, off, off, algo=algo)
        got_i, got_d = pointops.knn_prefix(ks, kw, wide_i, wide_d, p, p, off, off, algo=algo)
        if algo == "auto":
            assert torch.equal(got_i, want_i) and torch.equal(got_d, want_d)    # reference order, bit for bit
        else:
            assert _rows_as_sets_equal(got_i, want_i, got_d, want_d)            # the reference's set; equal distances in any order


def test_prefix_with_separate_queries_and_short_clouds():
    from contrastboundary_amd import pointops
    rng = np.random.default_rng(0)
    xyz = dev(rng.uniform(size=(5000, 3)).astype(np.float32)); q = dev(rng.uniform(size=(700, 3)).astype(np.float32))
    off = dev(np.int32([10, 2600, 5000])); qoff = dev(np.int32([100, 400, 700]))   # first cloud: 10 supports < K
    wide_i, wide_d = pointops.knnquery_raw(24, xyz, q, off, qoff)
    want_i, want_d = pointops.knnquery_raw(12, xyz, q, off, qoff)
    got_i, got_d = pointops.knn_prefix(12, 24, wide_i, wide_d, xyz, q, off, qoff)
    assert torch.equal(got_i, want_i) and torch.equal(got_d, want_d)


def test_cache_hint_runs_the_wide_search_once_and_serves_the_narrow_ones():
    from contrastboundary_amd import hotpath, pointops
    sc = hotpath.Scene.synthetic(16384, 32, seed=2)
    ref16 = pointops.knnquery_raw(16, sc.xyz, sc.xyz, sc.offset, sc.offset)
    ref8 = pointops.knnquery_raw(8, sc.xyz, sc.xyz, sc.offset, sc.offset)
    ref36 = pointops.knnquery_raw(36, sc.xyz, sc.xyz, sc.offset, sc.offset, algo="set")
    with pointops.neighbor_cache() as nc:
        nc.hint(sc.xyz, 36, "set")
        a = pointops.knnquery_raw(16, sc.xyz, sc.xyz, sc.offset, sc.offset)
        b = pointops.knnquery_raw(8, sc.xyz, sc.xyz, sc.offset, sc.offset)
        c = pointops.knnquery_raw(36, sc.xyz, sc.xyz, sc.offset, sc.offset, algo="set")
        assert nc.derived == 2 and nc.hits == 1 and nc.misses == 2
    assert torch.equal(a[0], ref16[0]) and torch.equal(a[1], ref16[1])
    assert torch.equal(b[0], ref8[0]) and torch.equal(b[1], ref8[1])
    assert _rows_as_sets_equal(c[0], ref36[0], c[1], ref36[1])
    # without a hint a wider result that is already in the cache is used all the same
    with pointops.neighbor_cache() as nc:
        pointops.knnquery_raw(36, sc.xyz, sc.xyz, sc.offset, sc.offset, algo="set")
        a2 = pointops.knnquery_raw(16, sc.xyz, sc.xyz, sc.offset, sc.offset)
        assert nc.derived == 1
    assert torch.equal(a2[0], ref16[0])


def test_hotpath_with_nested_searches_equals_the_plain_step():
    from contrastboundary_amd import hotpath
    sc = hotpath.Scene.synthetic(8192, 32, seed=3)
    st = hotpath.stages(sc, 16)
    ref = hotpath.run_once(sc, 16, {})
    sched = hotpath.Schedule(st, overlap=True, hints=hotpath.search_hints(sc))
    state = {}
    for _ in range(2):
        sched.run(state)
    torch.cuda.synchronize()
    assert torch.equal(state["idx"], ref["idx"]) and torch.equal(state["dist2"], ref["dist2"])
    assert torch.equal(torch.sort(state["cbl_idx"], 1)[0], torch.sort(ref["cbl_idx"], 1)[0])
    assert torch.equal(state["grouped"], ref["grouped"]) and torch.equal(state["kpconv"], ref["kpconv"])
    assert abs(float(state["cbl_loss"].detach()) - float(ref["cbl_loss"].detach())) <= 1e-5 * abs(float(ref["cbl_loss"].detach()))


@pytest.mark.parametrize("with_event", [False, True])
@pytest.mark.parametrize("kind", ["room", "lattice", "duplicates", "short"])
def test_nested_call_equals_the_two_searches(kind, with_event):
    """cbl_knnquery_nested (wide search + derivation in one call, optional event behind the wide part) on tie-heavy and short clouds"""
    from contrastboundary_amd import pointops, synthetic as S
    rng = np.random.default_rng(11)
    if kind == "room":
        xyz = S.s_room(20000, seed=4)[0]; offs = [7000, len(xyz)]
    elif kind == "lattice":
        g = np.stack(np.meshgrid(np.arange(15), np.arange(15), np.arange(15), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * 0.1
        xyz = g[rng.permutation(len(g))]; offs = [len(xyz)]
    elif kind == "duplicates":
        base = rng.uniform(size=(3000, 3)).astype(np.float32)
        xyz = np.concatenate([base, base[:1500]])[rng.permutation(4500)]; offs = [1500, 4500]
    else:                                                                      # clouds with fewer supports than K' / than K
        xyz = rng.uniform(size=(6000, 3)).astype(np.float32); offs = [10, 30, 6000]
    p = dev(xyz); off = dev(np.int32(offs))
    for ks, kw, algo_w, algo in ((16, 36, "set", "auto"), (8, 16, "auto", "auto"), (12, 40, "anytie", "set")):
        want_w = pointops.knnquery_raw(kw, p, p, off, off, algo=algo_w)
        want = pointops.knnquery_raw(ks, p, p, off, off, algo=algo)
        ev = None
        if with_event:
            ev = torch.cuda.Event(); ev.record()
        got = pointops._knnquery_nested(kw, algo_w, ks, algo, p, p, off, off, ev)
        assert got is not None
        wi, wd, gi, gd = got
        if with_event:                                                         # the wide result alone is complete behind the event
            other = torch.cuda.Stream()
            other.wait_event(ev)
            with torch.cuda.stream(other):
                wi_seen, wd_seen = wi.clone(), wd.clone()
            other.synchronize()
        torch.cuda.synchronize()
        if with_event:
            assert torch.equal(wi_seen, wi) and torch.equal(wd_seen, wd)
        if algo == "auto":
            assert torch.equal(gi, want[0]) and torch.equal(gd, want[1])
        else:
            assert _rows_as_sets_equal(gi, want[0], gd, want[1])
        if algo_w == "auto":
            assert torch.equal(wi, want_w[0]) and torch.equal(wd, want_w[1])
        elif algo_w == "set":
            assert _rows_as_sets_equal(wi, want_w[0], wd, want_w[1])
        else:
            assert torch.equal(wd, want_w[1])
