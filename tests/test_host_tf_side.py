"""Host logic of the TF-side mirrors that needs no GPU: the sample-string parser of the contrast head against the restatement, the deferred width trim of
the pyramid builder, the cached cumulative offsets, the scoping of the capture flag."""
import numpy as np
import pytest
import torch


def test_tf_sample_columns_agree_with_the_restatement():
    """heads.tf_sample_columns (what the kernel is handed: concatenated columns, one role per column, the 'R' reject mask) against
    cbl_oracle.tf_samples (head.py:551-625) on CPU tensors"""
    from contrastboundary_amd import heads
    from oracle import cbl_oracle as C
    rng = np.random.default_rng(0)
    m, k = 300, 11
    nbr = np.concatenate([np.arange(m)[:, None], rng.integers(0, m + 1, (m, k - 1))], 1).astype(np.int32)      # m = shadow
    hard = rng.integers(0, 4, m); hard[::13] = -1
    r1, r2 = rng.integers(0, m, (m, 6)).astype(np.int32), rng.integers(0, m, (m, 3)).astype(np.int32)
    r2[:, 0] = nbr[:, 2]                                                    # draws that are neighbours (or the shadow index)
    r2 = np.minimum(r2, m - 1)
    sample = "nn2-label-rand6-rand3R"
    samples, roles, valid = heads.tf_sample_columns(torch.from_numpy(nbr), sample, rand_idx=[torch.from_numpy(r1), torch.from_numpy(r2)])
    idx, pos, neg = C.tf_samples(hard, nbr, m, sample, [r1, r2])
    assert samples.dtype == torch.int32 and samples.shape == (m, 1 + idx.shape[1])
    np.testing.assert_array_equal(samples[:, 0].numpy(), nbr[:, 0])        # the self column stays in front (the kernel drops column 0)
    np.testing.assert_array_equal(samples[:, 1:].numpy(), idx)
    assert roles.tolist() == [heads.ROLE_POS] * 2 + [heads.ROLE_LABEL] * (k - 1) + [heads.ROLE_NEG] * 6 + [heads.ROLE_NEG_REJECT] * 3
    # what the kernel derives from roles + valid + labels is the restatement's positive / negative masks
    lab = np.concatenate([hard, [-1]])[np.minimum(idx, m)]
    ok = (lab >= 0) & (hard[:, None] >= 0)
    role = np.asarray(roles.tolist())[None, :]
    v = valid.numpy().astype(bool)
    k_nb = np.where(role == heads.ROLE_LABEL, ok, np.where(role == heads.ROLE_NEG_REJECT, v, True))
    k_pos = k_nb & np.where(role == heads.ROLE_LABEL, lab == hard[:, None], role == heads.ROLE_POS)
    np.testing.assert_array_equal(k_pos, pos); np.testing.assert_array_equal(k_nb & ~k_pos, neg)
    with pytest.raises(NotImplementedError):
        heads.tf_sample_columns(torch.from_numpy(nbr), "label-farthest4")
    with pytest.raises(ValueError):
        heads.tf_sample_columns(torch.from_numpy(nbr), "nn40")


def test_internal_draws_stay_inside_their_cloud():
    from contrastboundary_amd import heads
    nbr = torch.arange(50, dtype=torch.int32)[:, None].repeat(1, 4)
    lens = torch.tensor([20, 30], dtype=torch.int32)
    g = torch.Generator().manual_seed(1)
    samples, roles, valid = heads.tf_sample_columns(nbr, "rand9", batches_len=lens, generator=g)
    d = samples[:, 1:].numpy()
    assert (d[:20] < 20).all() and (d[20:] >= 20).all() and (d < 50).all() and valid is None and roles.tolist() == [heads.ROLE_NEG] * 9
    with pytest.raises(ValueError):
        heads.tf_sample_columns(nbr, "rand4", batches_len=torch.tensor([20, 31], dtype=torch.int32))


def test_deferred_widths_are_trimmed_like_the_reference_slices():
    from contrastboundary_amd import tf_ops
    a, b, c = torch.arange(12, dtype=torch.int32).view(3, 4), torch.arange(15, dtype=torch.int32).view(3, 5), torch.zeros((0, 6), dtype=torch.int32)
    got = tf_ops.trim_neighbor_widths([(a, torch.tensor([9], dtype=torch.int32)), (b, torch.tensor([3], dtype=torch.int32)), (c, torch.tensor([0], dtype=torch.int32))])
    assert got[0] is a                                                     # largest neighbourhood >= limit: the table as it is
    assert got[1].shape == (3, 3) and got[1].is_contiguous() and torch.equal(got[1], b[:, :3])
    assert got[2].shape == (0, 0)
    assert tf_ops.trim_neighbor_widths([]) == []


def test_cumulative_offsets_are_cached_per_tensor_and_version():
    """only while a pyramid is built op by op (a lengths buffer rewritten through the C ABI bumps no version counter, so nothing is remembered
    outside the builder, whose vectors are written once); inside: per tensor object and version"""
    from contrastboundary_amd import tf_ops
    lens = torch.tensor([3, 4, 5], dtype=torch.int32)
    o0 = tf_ops._offsets(lens)
    assert o0.tolist() == [3, 7, 12] and tf_ops._offsets(lens) is not o0      # outside the builder: recomputed every time
    with tf_ops._offsets_cached():
        o1 = tf_ops._offsets(lens)
        assert o1.tolist() == [3, 7, 12] and tf_ops._offsets(lens) is o1
        lens[1] = 6                                                        # in-place edit bumps the version: recomputed
        assert tf_ops._offsets(lens).tolist() == [3, 9, 14]
        other = torch.tensor([3, 6, 5], dtype=torch.int32)
        assert tf_ops._offsets(other) is not tf_ops._offsets(lens)
    assert not tf_ops._offset_state.cache                                     # dropped with the block


def test_offset_cache_is_per_thread():
    """the pyramid loader builds on a thread of its own: its block must neither switch the main thread's caching on nor clear what the main
    thread's block remembers"""
    import threading
    from contrastboundary_amd import tf_ops
    lens = torch.tensor([2, 2], dtype=torch.int32)
    seen = {}

    def worker():
        seen["on_before"] = getattr(tf_ops._offset_state, "on", 0)
        with tf_ops._offsets_cached():
            a = tf_ops._offsets(lens)
            seen["cached_inside"] = tf_ops._offsets(lens) is a
        seen["on_after"] = tf_ops._offset_state.on

    with tf_ops._offsets_cached():
        mine = tf_ops._offsets(lens)
        t = threading.Thread(target=worker); t.start(); t.join()
        assert tf_ops._offsets(lens) is mine                                # the worker's exit did not clear this thread's cache
    assert seen == {"on_before": 0, "cached_inside": True, "on_after": 0}
    assert tf_ops._offsets(lens) is not mine                                # and outside every block nothing is remembered


def test_capture_flag_is_scoped_to_its_streams():
    """streams_ordered_by_caller answers per stream handle (a process-wide flag silenced other threads' waits)"""
    from contrastboundary_amd import neighbor_state as NS

    class S:                                                               # the class only reads .cuda_stream
        def __init__(self, h): self.cuda_stream = h
    a, b, c = S(11), S(12), S(13)
    assert not NS.streams_ordered_by_caller.applies(a)
    with NS.streams_ordered_by_caller([a, b]):
        assert NS.streams_ordered_by_caller.applies(a) and NS.streams_ordered_by_caller.applies(b) and not NS.streams_ordered_by_caller.applies(c)
        with NS.streams_ordered_by_caller([b, c]):
            assert NS.streams_ordered_by_caller.applies(c)
        assert NS.streams_ordered_by_caller.applies(b) and not NS.streams_ordered_by_caller.applies(c)
    assert not NS.streams_ordered_by_caller.applies(a) and not NS.streams_ordered_by_caller.applies(b)
