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


def test_time_crops(coord, voxel_max, potentials):
    """the spatially regular crops of the test loop, /root/reference/pytorch/tool/test.py:197-215: until every point is covered, take the
    point of minimum potential as centre, crop its voxel_max nearest points (stable order where the reference's quicksort argsort is
    free) and raise the potentials of the cropped points by (1 - d2 / max d2)^2.  `potentials` = the reference's np.random.rand(n) * 1e-3.
    -> list of index arrays (ascending distance)"""
    coord = np.asarray(coord)
    pot = np.array(potentials, dtype=np.float64)
    n = coord.shape[0]
    covered = np.zeros(n, bool)
    crops = []
    while covered.sum() != n:                                        # idx_uni.size != idx_part.shape[0], :200
        init = int(np.argmin(pot))                                   # :202
        d = coord - coord[init]
        dist = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]      # np.sum(np.power(.., 2), 1), :204
        idx_crop = np.argsort(dist.astype(np.float64), kind="stable")[:voxel_max]   # :205
        dc = dist[idx_crop]
        delta = np.square(1 - dc / np.max(dc))                       # :209
        pot[idx_crop] += delta                                       # :210
        covered[idx_crop] = True                                     # np.unique of the concatenated crops, :216
        crops.append(idx_crop)
    return crops
