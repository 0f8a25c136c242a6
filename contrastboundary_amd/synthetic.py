"""Deterministic synthetic scenes: no dataset ships with the reference and there is no network.

S-uniform(N): uniform points in a cube sized so that a 0.1 m ball holds ~20 points.
S-room(N):    points on the faces of a 6x4x3 m room and of 8 random axis-aligned cuboids inside it, 5 mm jitter,
              de-duplicated on a 0.04 m voxel grid (one point per voxel, like data_prepare's voxelize,
              /root/reference/pytorch/util/data_util.py:45-67), shuffled, coord -= min; label = face/cuboid id
              mod 13 (blocky labels -> a realistic fraction of boundary points for the CBL head).
"""
import numpy as np


def s_uniform(n, seed=0):
    rng = np.random.default_rng(seed)
    side = (n / (20.0 / (4.0 / 3.0 * np.pi * 0.1 ** 3))) ** (1.0 / 3.0)
    return rng.uniform(0.0, side, (n, 3)).astype(np.float32)


def _box_faces(rng, lo, hi, count, label0):
    """`count` points on the 6 faces of the box [lo,hi], area-weighted; labels label0..label0+5"""
    ext = hi - lo
    areas = np.array([ext[1] * ext[2], ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[2], ext[0] * ext[1], ext[0] * ext[1]])
    face = rng.choice(6, size=count, p=areas / areas.sum())
    pts = lo + rng.uniform(size=(count, 3)) * ext
    axis = face // 2
    side = face % 2
    pts[np.arange(count), axis] = np.where(side == 0, lo[axis], hi[axis])
    return pts, label0 + face


def s_room(n, seed=0, scale=1.0, voxel=0.04, num_classes=13):
    """-> xyz (n,3) f32, labels (n,) int64 in [0,num_classes)"""
    rng = np.random.default_rng(seed)
    room_lo, room_hi = np.zeros(3), np.array([6.0, 4.0, 3.0]) * np.array([scale, scale, 1.0])
    boxes = [(room_lo, room_hi)]
    for _ in range(int(8 * scale * scale)):
        size = rng.uniform(0.4, 1.6, 3)
        lo = rng.uniform(room_lo, room_hi - size)
        boxes.append((lo, lo + size))
    areas = np.array([2 * ((h - l)[0] * (h - l)[1] + (h - l)[0] * (h - l)[2] + (h - l)[1] * (h - l)[2]) for l, h in boxes])
    if n > 1.2 * areas.sum() / (voxel * voxel):                     # more points than the surfaces have voxels: fail now, not after 8 doublings
        raise ValueError(f"s_room: {n} points do not fit a room of scale {scale} at voxel {voxel}; raise `scale` (~sqrt(n / 40000))")
    out_p, out_l = [], []
    want = int(n * 2.2) + 1000
    for trial in range(8):
        counts = rng.multinomial(want, areas / areas.sum())
        ps, ls = [], []
        for bi, ((lo, hi), cnt) in enumerate(zip(boxes, counts)):
            p, l = _box_faces(rng, lo, hi, cnt, 6 * bi)
            ps.append(p); ls.append(l)
        p = np.concatenate(ps) + rng.normal(0, 0.005, (want, 3))
        l = np.concatenate(ls)
        key = np.floor(p / voxel).astype(np.int64)
        _, first = np.unique(key, axis=0, return_index=True)
        first = rng.permutation(first)
        out_p, out_l = p[first], l[first]
        if len(first) >= n:
            break
        want *= 2
    assert len(out_p) >= n, "room too small for the requested point count at this voxel size"
    xyz = out_p[:n]
    xyz = (xyz - xyz.min(0)).astype(np.float32)
    return xyz, (out_l[:n] % num_classes).astype(np.int64)


def offsets(n, b, seed=0, jitter=0.1):
    """cumulative end offsets of b clouds covering n rows, lengths n/b +-jitter"""
    if b == 1:
        return np.array([n], np.int32)
    rng = np.random.default_rng(seed)
    lens = (n / b * (1 + rng.uniform(-jitter, jitter, b))).astype(np.int64)
    lens[-1] = n - lens[:-1].sum()
    return np.cumsum(lens).astype(np.int32)
