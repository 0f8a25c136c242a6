"""AdaptiveWeight / PosPool forward+backward at the ConvNet layer shapes (N=200k pyramid)."""
import numpy as np, torch
from contrastboundary_amd import local_aggregation as LA, synthetic as S, tf_ops
def timeit(fn, reps=11):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
xyz = torch.from_numpy(S.s_room(200000, 0, scale=4.0)[0]).cuda(); lens = torch.tensor([200000], dtype=torch.int32, device="cuda")
pyr = tf_ops.segmentation_inputs_radius(xyz, lens, 0.04, 5.0, 5, [26, 31, 38, 41, 39])
for l, C in enumerate([72, 144, 288, 576, 1152]):
    q = pyr["points"][l]; nb = pyr["neighbors"][l].contiguous(); n, K = nb.shape
    f = torch.randn(n, C, device="cuda").requires_grad_(True); W = torch.randn(3, C, device="cuda"); b = torch.randn(C, device="cuda")
    r = 0.1 * 2 ** l
    go = torch.randn(n, C, device="cuda")
    gath = n * K * C * 4
    t_aw = timeit(lambda: LA.adaptive_weight(q, q, nb, f, r, W, b, "mean"))
    def fb():
        f.grad = None; LA.adaptive_weight(q, q, nb, f, r, W, b, "mean").backward(go)
    t_awfb = timeit(fb)
    t_pp = timeit(lambda: LA.pospool(q, q, nb, f, r, "sin_cos", "mean"))
    t_px = timeit(lambda: LA.pospool(q, q, nb, f, r, "xyz", "mean"))
    def pb():
        f.grad = None; LA.pospool(q, q, nb, f, r, "sin_cos", "mean").backward(go)
    t_ppfb = timeit(pb)
    print(f"layer {l} n={n} K={K} C={C}: AW fwd {t_aw:7.1f} us ({gath/t_aw/1e6:5.2f} TB/s gathered) fwd+bwd {t_awfb:7.1f} | PosPool sin_cos fwd {t_pp:7.1f} fwd+bwd {t_ppfb:7.1f} | xyz fwd {t_px:7.1f}")
