"""The measured hot path (BASELINE.json metric): one pass over one S3DIS-shaped scene of
    KNN (K=16)  ->  neighbour grouping (xyz-centred + features, (N,K,3+C))  ->  local aggregation over the K
    neighbours  ->  CBL head (neighbour search + pair mining + loss, forward and backward w.r.t. the features)
Each stage is one or a few C-ABI launches on the current stream; `stages()` lists them with the ALGORITHMIC bytes
/ flops of SURVEY.md §8(d) so bench.py can turn a measured duration into a roofline fraction.
"""
import torch

from . import pointops


class Scene:
    """device-resident synthetic scene: xyz (N,3), feat (N,C), labels (N,), offset (b,)"""

    def __init__(self, xyz, feat, labels, offset):
        self.xyz, self.feat, self.labels, self.offset = xyz, feat, labels, offset
        self.n, self.c = feat.shape

    @staticmethod
    def synthetic(n, c, seed=0, b=1, device="cuda"):
        import numpy as np
        from . import synthetic as S
        xyz, labels = S.s_room(n, seed)
        rng = np.random.default_rng(seed + 1000)
        feat = rng.normal(size=(n, c)).astype(np.float32)
        off = S.offsets(n, b, seed)
        t = lambda a: torch.from_numpy(a).to(device)
        return Scene(t(xyz), t(feat), t(labels), t(off))


def stages(scene, k=16):
    """-> list of (name, fn(state) -> None, algorithmic_bytes, algorithmic_flops); fns communicate through `state`"""
    n, c = scene.n, scene.c
    st = []

    def knn(s):
        s["idx"], s["dist2"] = pointops.knnquery_raw(k, scene.xyz, scene.xyz, scene.offset, scene.offset)
    # compulsory 12n + 12m + 8mK bytes
    st.append(("knnquery_k%d" % k, knn, 12 * n + 12 * n + 8 * n * k, 8.0 * n * n))

    def group(s):
        s["grouped"] = pointops.queryandgroup(k, scene.xyz, scene.xyz, scene.feat, s["idx"], scene.offset, scene.offset, use_xyz=True)
    # a3 fused queryandgroup: 4mK + 12n + 12m + 4nC + 4mK(3+C)
    st.append(("queryandgroup", group, 4 * n * k + 12 * n + 12 * n + 4 * n * c + 4 * n * k * (3 + c), 3.0 * n * k))
    return st


def run_once(scene, k=16, state=None):
    state = {} if state is None else state
    for _, fn, _, _ in stages(scene, k):
        fn(state)
    return state
