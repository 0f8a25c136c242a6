"""one PointTransformerLayer forward+backward at a stage shape, for rocprofv3: python layer_prof.py <n> <K> <C>"""
import sys, torch, time
from contrastboundary_amd import blocks, synthetic as S, pointops
n, K, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
xyz = torch.from_numpy(S.s_room(n, seed=0)[0]).cuda(); o = torch.tensor([n], dtype=torch.int32, device="cuda")
layer = blocks.PointTransformerLayer(C, C, 8, K).cuda().train()
x = torch.randn(n, C, device="cuda", requires_grad=True); g = torch.randn(n, C, device="cuda")
idx, _ = pointops.knnquery(K, xyz, xyz, o, o)
def step():
    x.grad = None
    y = layer([xyz, x, o], idx)
    (y * g).sum().backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); print(f"n={n} K={K} C={C}: {(time.perf_counter()-t0)/10*1e3:.3f} ms per layer fwd+bwd")
