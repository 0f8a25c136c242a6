import torch
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mb in (175, 700, 2800):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    t_fill = timeit(lambda: x.fill_(1.0)); t_copy = timeit(lambda: y.copy_(x)); t_read = timeit(lambda: x.sum())
    print(f"{mb} MB: fill {t_fill:7.1f} us = {mb*1.048576/t_fill*1e3/1e3:5.2f} TB/s write | copy {t_copy:7.1f} us = {2*mb*1.048576/t_copy:5.2f} TB/s r+w | sum {t_read:7.1f} us = {mb*1.048576/t_read:5.2f} TB/s read")
