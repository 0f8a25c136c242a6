"""How much do the gather kernels gain if the scene is in spatial (Morton) order instead of the generator's shuffled order?"""
import numpy as np, torch
from contrastboundary_amd import pointops, hotpath, local_aggregation as LA, heads
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def spread(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x030000ff; v = (v | (v << 8)) & 0x0300f00f; v = (v | (v << 4)) & 0x030c30c3; v = (v | (v << 2)) & 0x09249249
    return v
a = hotpath.Scene.synthetic_numpy(40960, 64, seed=0)
for name in ("shuffled", "morton"):
    xyz = a["xyz"]
    if name == "morton":
        q = np.clip(((xyz - xyz.min(0)) * (1023.99 / (xyz.max(0) - xyz.min(0)).max())).astype(np.int64), 0, 1023)
        perm = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")
    else:
        perm = np.arange(40960)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    X, F, Lb, Lat = t(xyz[perm]), t(a["feat"][perm]), t(a["labels"][perm]), t(a["latent"][perm])
    o = t(a["offset"]); kp, kw = t(a["kernel_points"]), t(a["kernel_weights"])
    idx, _ = pointops.knnquery_raw(16, X, X, o, o)
    nidx, _ = pointops.knnquery_raw(36, X, X, o, o, algo="set")
    lat = Lat.clone().requires_grad_(True)
    def cbl():
        lat.grad = None; heads.point_contrast(lat, Lb, nidx, 1.0, 0.1).backward()
    print(f"{name:9s} knn16 {timeit(lambda: pointops.knnquery_raw(16, X, X, o, o)):6.1f}  knn36set {timeit(lambda: pointops.knnquery_raw(36, X, X, o, o, algo='set')):6.1f}  "
          f"queryandgroup {timeit(lambda: pointops.queryandgroup(16, X, X, F, idx, o, o)):6.1f}  kpconv {timeit(lambda: LA.kpconv(X, X, idx, F, kp, kw, 0.12)):6.1f}  "
          f"cbl fwd+bwd {timeit(cbl):6.1f}  grouping {timeit(lambda: pointops.grouping(F, idx)):6.1f} us")
