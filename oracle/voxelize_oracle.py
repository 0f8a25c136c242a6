"""voxelize_oracle.py — TEST INFRASTRUCTURE ONLY.
numpy restatement of /root/reference/pytorch/util/voxelize.py:4-16,38-56 with a STABLE argsort (the reference's default
quicksort leaves the order inside a voxel unspecified).  Pinned by tests/golden/voxelize.npz, produced by importing and running
the reference module itself (tests/golden/gen_voxelize_goldens.py): keys and counts must match exactly, idx_sort per voxel as a set."""
import numpy as np


def fnv_hash_vec(arr):
    arr = np.asarray(arr).astype(np.uint64)
    h = np.uint64(14695981039346656037) * np.ones(arr.shape[0], dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(arr.shape[1]):
            h = h * np.uint64(1099511628211)
            h = np.bitwise_xor(h, arr[:, j])
    return h


def voxelize(coord, voxel_size):
    """-> key (n,), idx_sort (n,) stable, start (v,), count (v,)"""
    coord = np.asarray(coord)
    disc = np.floor(coord / coord.dtype.type(voxel_size))
    key = fnv_hash_vec(disc)
    idx_sort = np.argsort(key, kind="stable")
    ks = key[idx_sort]
    _, start, count = np.unique(ks, return_index=True, return_counts=True)
    return key, idx_sort, start, count


def crop_order(coord, center):
    coord = np.asarray(coord)
    d = coord - coord[center]
    d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float64)
    return np.argsort(d2, kind="stable")
