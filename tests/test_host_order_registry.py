"""Host-side bookkeeping of the processing orders (contrastboundary_amd/pointops.py): no GPU, the streams are stand-ins."""
import torch


class _Stream:
    def __init__(self, handle):
        self.cuda_stream = handle


def test_a_cache_drops_the_orders_registered_during_its_pass():
    from contrastboundary_amd import pointops
    pointops._order_registry.clear()
    pts = torch.zeros(pointops.ORDER_MIN_POINTS, 3)
    idx = torch.zeros(pointops.ORDER_MIN_POINTS, 4, dtype=torch.int32)
    order = torch.arange(pointops.ORDER_MIN_POINTS, dtype=torch.int32)
    s = _Stream(7)
    with pointops.neighbor_cache():
        assert pointops._order_wanted(pts, s.cuda_stream)
        pointops._order_register(pts, order, s)
        pointops._order_alias(idx, pts)
        assert not pointops._order_wanted(pts, s.cuda_stream)          # produced once per pass and stream
        assert pointops._order_wanted(pts, 8)                          # another stream still has to be ordered behind the producer
        assert len(pointops._order_registry) == 2
    assert len(pointops._order_registry) == 0                          # nothing is carried to the next pass
    assert pointops._order_wanted(pts, s.cuda_stream)


def test_orders_outside_a_cache_stay_until_evicted():
    from contrastboundary_amd import pointops
    pointops._order_registry.clear()
    s = _Stream(3)
    keep = []
    for i in range(pointops._ORDER_REGISTRY_MAX + 5):
        pts = torch.zeros(pointops.ORDER_MIN_POINTS + i, 3)
        keep.append(pts)
        pointops._order_register(pts, torch.zeros(pts.shape[0], dtype=torch.int32), s)
    assert len(pointops._order_registry) == pointops._ORDER_REGISTRY_MAX
    assert pointops._order_wanted(keep[0], s.cuda_stream)              # the oldest entries were evicted
    assert not pointops._order_wanted(keep[-1], s.cuda_stream)
    pointops._order_registry.clear()


def test_small_clouds_never_ask_for_an_order():
    from contrastboundary_amd import pointops
    pts = torch.zeros(pointops.ORDER_MIN_POINTS - 1, 3)
    assert not pointops._order_wanted(pts, 1)


def test_a_kept_cache_keeps_its_orders_alive_past_their_eviction():
    """the registry is a bounded LRU; a cache that outlives its pass (keep = True: geometry.StaticGeometry, whose orders' addresses are baked
    into captured hipGraphs) must hold the tensors itself — they were once freed under a replaying graph when later passes filled the registry"""
    import weakref
    from contrastboundary_amd import pointops
    pointops._order_registry.clear()
    s = _Stream(5)
    cache = pointops.neighbor_cache()
    cache.keep = True
    pts = torch.zeros(pointops.ORDER_MIN_POINTS, 3)
    with cache:
        order = torch.arange(pointops.ORDER_MIN_POINTS, dtype=torch.int32)
        pointops._order_register(pts, order, s)
        alive = weakref.ref(order)
        del order
    assert len(pointops._order_registry) == 1                          # kept: still registered after the pass
    keep = []
    for i in range(pointops._ORDER_REGISTRY_MAX + 2):                  # later passes push it out of the registry ...
        q = torch.zeros(pointops.ORDER_MIN_POINTS + 1 + i, 3)
        keep.append(q)
        pointops._order_register(q, torch.zeros(q.shape[0], dtype=torch.int32), s)
    assert pointops._order_wanted(pts, s.cuda_stream)
    assert alive() is not None                                         # ... the tensor itself lives as long as the cache does
    del cache
    import gc; gc.collect()
    assert alive() is None
    pointops._order_registry.clear()
