import ctypes, torch, numpy as np
from contrastboundary_amd import pointops, synthetic as S, _lib
n = 40960
xyz = torch.from_numpy(S.s_room(n, seed=0)[0]).cuda()
o = torch.tensor([n], dtype=torch.int32, device="cuda"); no = torch.tensor([n // 4], dtype=torch.int32, device="cuda")
pointops.furthestsampling(xyz, o, no); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
print("rc", _lib.lib().cbl_debug_fps_prof(out))
names = ["S1", "barA", "S2", "barB", "S3a", "barC", "S3b", "qn_total"]
for w in (0, 1):
    v = np.array(out[8 * w: 8 * w + 8], dtype=np.float64)
    print("wave", 0 if w == 0 else 5, {k: round(x / 10239, 1) for k, x in zip(names, v)}, "sum cycles/sample", round(v[:7].sum() / 10239, 1))
