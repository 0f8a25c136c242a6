"""GPU microbench: per-kernel time and algorithmic GB/s at C2 (N=40960, K=16, C=64)."""
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

sc = hotpath.Scene.synthetic(40960, 64, seed=0)
n, K, C = 40960, 16, 64
idx, _ = pointops.knnquery_raw(K, sc.xyz, sc.xyz, sc.offset, sc.offset)
MB = 1e6
def rep(name, us, bytes_):
    print(f"{name:34s} {us:8.1f} us   {bytes_/us/1e3:8.1f} GB/s algorithmic   {bytes_/us/1e3/8000*100:5.1f}% of 8 TB/s")
rep("grouping fwd (K3, C=64)", timeit(lambda: pointops.grouping(sc.feat, idx)), 4*n*K + 4*n*C + 4*n*K*C)
g = torch.ones(n, K, C, device="cuda"); f = sc.feat.clone().requires_grad_(True)
def gb():
    f.grad = None; pointops.grouping(f, idx).backward(g)
rep("grouping fwd+bwd (K3+K4)", timeit(gb), 2*(4*n*K + 4*n*C + 4*n*K*C))
rep("queryandgroup (3+C)", timeit(lambda: pointops.queryandgroup(K, sc.xyz, sc.xyz, sc.feat, idx, sc.offset, sc.offset)), 4*n*K + 24*n + 4*n*C + 4*n*K*(3+C))
rep("knnquery K=16 (auto)", timeit(lambda: pointops.knnquery_raw(16, sc.xyz, sc.xyz, sc.offset, sc.offset)), 24*n + 8*n*16)
rep("knnquery K=36 (auto)", timeit(lambda: pointops.knnquery_raw(36, sc.xyz, sc.xyz, sc.offset, sc.offset)), 24*n + 8*n*36)
rep("knnquery K=8  (auto)", timeit(lambda: pointops.knnquery_raw(8, sc.xyz, sc.xyz, sc.offset, sc.offset)), 24*n + 8*n*8)
rep("kpconv fwd (KP=15)", timeit(lambda: LA.kpconv(sc.xyz, sc.xyz, idx, sc.feat, sc.kernel_points, sc.kernel_weights, 0.12)), 24*n + 4*n*C*2 + 4*n*K)
pos = torch.randn(n, K, C, device="cuda"); w = torch.randn(n, K, 8, device="cuda")
rep("aggregation fwd (K9, w_c=8)", timeit(lambda: pointops.aggregation(sc.feat, pos, w, idx)), 4*n*K + 4*n*C + 4*n*K*C + 4*n*K*8 + 4*n*C)
rep("subtraction fwd (K7)", timeit(lambda: pointops.subtraction(sc.feat, sc.feat, idx)), 4*n*K + 8*n*C + 4*n*K*C)
m = n // 4
noff = torch.tensor([m], dtype=torch.int32, device="cuda")
print(f"{'furthestsampling 40960 -> 10240':34s} {timeit(lambda: pointops.furthestsampling(sc.xyz, sc.offset, noff), 3):8.1f} us")
xyz1 = sc.xyz[:10240].contiguous(); o1 = torch.tensor([10240], dtype=torch.int32, device="cuda"); n1 = torch.tensor([2560], dtype=torch.int32, device="cuda")
print(f"{'furthestsampling 10240 -> 2560':34s} {timeit(lambda: pointops.furthestsampling(xyz1, o1, n1), 3):8.1f} us")
nidx, _ = pointops.knnquery_raw(36, sc.xyz, sc.xyz, sc.offset, sc.offset)
lat = sc.latent.clone().requires_grad_(True)
def cbl():
    lat.grad = None; heads.point_contrast(lat, sc.labels, nidx, 1.0, 0.1).backward()
rep("CBL mining+loss fwd+bwd (K=36,d=32)", timeit(cbl), 2*(4*n*35 + 4*n*32 + 4*n) + 4*n*32)
x = torch.empty(64*1024*1024, device="cuda"); y = torch.empty_like(x)
rep("torch d2d copy 256 MB (ceiling)", timeit(lambda: y.copy_(x)), 2*x.numel()*4)
# ---- TF-side ops at C5 scale (N = 200000)
from contrastboundary_amd import tf_ops, voxelize as VZ, synthetic as S
xyz5, _ = S.s_room(200000, seed=0, scale=4.0)
x5 = torch.from_numpy(xyz5).cuda(); l5 = torch.tensor([200000], dtype=torch.int32, device="cuda")
print(f"{'radius r=0.1 limit 26 (N=200k)':34s} {timeit(lambda: tf_ops.tf_batch_neighbors(x5, x5, l5, l5, 0.1, 26, exact_shape=False), 10):8.1f} us")
print(f"{'grid subsample dl=0.08 (N=200k)':34s} {timeit(lambda: tf_ops.tf_batch_subsampling(x5, l5, 0.08), 10):8.1f} us  (incl. 1 host sync)")
print(f"{'pyramid 5 layers (N=200k)':34s} {timeit(lambda: tf_ops.segmentation_inputs_radius(x5, l5, 0.04, 5.0, 5, [26, 31, 38, 41, 39]), 5):8.1f} us")
print(f"{'voxelize 0.04 (N=200k)':34s} {timeit(lambda: VZ.voxelize(x5, 0.04, mode=1), 10):8.1f} us  (incl. 1 host sync)")
print(f"{'knnquery K=16 (N=200k)':34s} {timeit(lambda: pointops.knnquery_raw(16, x5, x5, torch.cumsum(l5,0,dtype=torch.int32), torch.cumsum(l5,0,dtype=torch.int32)), 10):8.1f} us")
