import torch, torch.nn.functional as F
from contrastboundary_amd import dense
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rows, cin, cout in [(40960, 32, 32), (40960, 6, 32), (10240, 64, 64), (10240, 32, 64), (327680, 32, 4), (163840, 64, 8), (163840, 3, 64), (163840, 8, 8), (40960, 64, 64), (40960, 128, 16)]:
    x = torch.randn(rows, cin, device="cuda", requires_grad=True); w = torch.randn(cout, cin, device="cuda", requires_grad=True); b = torch.randn(cout, device="cuda", requires_grad=True)
    g = torch.randn(rows, cout, device="cuda")
    def run(fn):
        def f():
            x.grad = w.grad = b.grad = None
            fn(x, w, b).backward(g)
        return f
    tf = timeit(lambda: F.linear(x, w, b)); td = timeit(lambda: dense._SkinnyLinear.apply(x, w, b))
    tfb = timeit(run(F.linear)); tdb = timeit(run(lambda x, w, b: dense._SkinnyLinear.apply(x, w, b)))
    print(f"rows={rows:7d} {cin:3d}->{cout:3d}: fwd torch {tf:7.1f} skinny {td:7.1f} | fwd+bwd torch {tfb:7.1f} skinny {tdb:7.1f} us")
