import torch, torch.nn.functional as F
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("preferred:", torch.backends.cuda.preferred_blas_library())
shapes = [(640, 256, 256), (2560, 128, 128), (160, 512, 512), (10240, 256, 32), (40960, 67, 128), (10240, 131, 256), (2560, 512, 64)]
for lib in ("default", "hipblaslt", "cublas"):
    try:
        if lib != "default": torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "unavailable", e); continue
    for rows, cin, cout in shapes:
        x = torch.randn(rows, cin, device="cuda"); w = torch.randn(cout, cin, device="cuda"); b = torch.randn(cout, device="cuda")
        g = torch.randn(rows, cout, device="cuda")
        t1 = timeit(lambda: F.linear(x, w, b)); t2 = timeit(lambda: g.t() @ x); t3 = timeit(lambda: g @ w)
        print(f"{lib:10s} ({rows},{cin},{cout}): fwd {t1:7.1f}  wgrad {t2:7.1f}  dgrad {t3:7.1f} us")
