"""Host-side logic of pointops.neighbor_cache (keys, hints, widest-result bookkeeping): plain CPU tensors stand in for the device ones,
no kernel is launched."""
import torch


def _geo(n=10):
    xyz = torch.zeros(n, 3); off = torch.tensor([n], dtype=torch.int32)
    return xyz, xyz, off, off


def test_identical_requests_hit_and_a_reference_order_result_serves_a_set_request():
    from contrastboundary_amd import pointops
    t = _geo()
    idx, d2 = torch.zeros(10, 4, dtype=torch.int32), torch.zeros(10, 4)
    with pointops.neighbor_cache() as nc:
        assert nc.lookup(4, "auto", t) is None
        nc.insert(4, "auto", t, idx, d2)
        hit = nc.lookup(4, "auto", t)
        assert hit[0] is idx and hit[1] is d2
        assert nc.lookup(4, "set", t)[0] is idx                     # reference order is also the reference set
        assert nc.lookup(5, "auto", t) is None                      # another neighbourhood size
        assert (nc.hits, nc.misses) == (2, 2)
    assert not nc.store                                             # dropped with the pass


def test_a_set_order_result_does_not_serve_a_reference_order_request():
    from contrastboundary_amd import pointops
    t = _geo()
    with pointops.neighbor_cache() as nc:
        nc.insert(4, "set", t, torch.zeros(10, 4, dtype=torch.int32), torch.zeros(10, 4))
        assert nc.lookup(4, "auto", t) is None


def test_keys_follow_storage_shape_and_version():
    from contrastboundary_amd import pointops
    xyz, _, off, _ = _geo()
    with pointops.neighbor_cache() as nc:
        nc.insert(4, "auto", (xyz, xyz, off, off), torch.zeros(10, 4, dtype=torch.int32), torch.zeros(10, 4))
        other = xyz.clone()
        assert nc.lookup(4, "auto", (other, other, off, off)) is None   # same values, other storage
        xyz.add_(1.0)                                                   # refreshed in place: version counter moved
        assert nc.lookup(4, "auto", (xyz, xyz, off, off)) is None
    with pointops.neighbor_cache() as nc:
        nc.ignore_version = True                                        # static geometry: the caller vouches for the contents
        nc.insert(4, "auto", (xyz, xyz, off, off), torch.zeros(10, 4, dtype=torch.int32), torch.zeros(10, 4))
        xyz.add_(1.0)
        assert nc.lookup(4, "auto", (xyz, xyz, off, off)) is not None


def test_widest_result_and_hints_per_geometry():
    from contrastboundary_amd import pointops
    t = _geo()
    with pointops.neighbor_cache() as nc:
        assert nc.wider(4, t) is None
        nc.insert(8, "set", t, torch.zeros(10, 8, dtype=torch.int32), torch.zeros(10, 8))
        nc.insert(6, "auto", t, torch.zeros(10, 6, dtype=torch.int32), torch.zeros(10, 6))
        w = nc.wider(4, t)
        assert w[0] == 8 and tuple(w[1].shape) == (10, 8)          # the widest one stored, not the latest
        assert nc.wider(8, t) is None                               # nothing wider than 8
        nc.insert(9, "anytie", t, torch.zeros(10, 9, dtype=torch.int32), torch.zeros(10, 9))
        assert nc.wider(8, t) is None                               # an any-tie result fixes no neighbour set: never a source
        nc.hint(t[0], 36, "set")
        assert nc.hints[nc._geo(t[0], t[1])] == (36, "set")
        other = torch.zeros(10, 3)
        assert nc._geo(other, other) not in nc.hints


def test_caches_nest_and_restore_the_outer_one():
    from contrastboundary_amd import pointops
    assert pointops.neighbor_cache.active() is None
    with pointops.neighbor_cache() as outer:
        with pointops.neighbor_cache() as inner:
            assert pointops.neighbor_cache.active() is inner
        assert pointops.neighbor_cache.active() is outer
    assert pointops.neighbor_cache.active() is None


def test_static_geometry_refuses_a_batch_with_other_cloud_boundaries():
    """StaticGeometry derives FPS counts / n_max / new_offset from the FIRST batch's cloud boundaries once (host_ends); a later batch with the same total but other
    boundaries must be refused where that is checkable without a device wait (host tensors, lists)"""
    import pytest
    import torch
    from contrastboundary_amd import geometry
    g = object.__new__(geometry.StaticGeometry)
    g.host_ends = [1000, 2500, 4096]
    g.check_offset(torch.tensor([1000, 2500, 4096], dtype=torch.int32))
    g.check_offset([1000, 2500, 4096])
    g.check_offset(None)
    with pytest.raises(ValueError):
        g.check_offset(torch.tensor([1200, 2500, 4096], dtype=torch.int32))
