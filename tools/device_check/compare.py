"""device output of tools/device_check/cbl_check.cpp against the host-emulated run of the same program:
    python tools/device_check/compare.py gpurun_out/cbl_check_dev.bin /tmp/cbl_check_host.bin"""
import sys

import numpy as np

N = 3000


def read(path):
    raw = np.fromfile(path, np.uint8)
    out, o = [], 0

    def take(count, dtype):
        nonlocal o
        a = raw[o:o + 4 * count].view(dtype); o += 4 * count
        return a
    while o < len(raw):
        nsample, d, flags, scale = take(4, np.int32)
        c = dict(key=(int(nsample), int(d), int(flags), int(scale)), loss=take(1, np.float32)[0], stats=take(2, np.float32), per_point=take(N, np.float32),
                 mask=take(N, np.int32), grad=take(N * int(d), np.float32))
        if flags <= 1:
            c["loss2"] = take(1, np.float32)[0]; c["grad2"] = take(N * int(d), np.float32)
        out.append(c)
    return out


a, b = read(sys.argv[1]), read(sys.argv[2])
assert len(a) == len(b), (len(a), len(b))
worst = 0.0
for x, y in zip(a, b):
    assert x["key"] == y["key"]
    assert np.array_equal(x["mask"], y["mask"]), x["key"]
    for k in ("grad", "grad2", "per_point"):
        if k in x:
            assert np.isfinite(x[k]).all(), (x["key"], k)
            s = max(float(np.abs(y[k]).max()), 1e-20)
            e = float(np.abs(x[k] - y[k]).max()) / s
            worst = max(worst, e)
            assert e < 1e-4, (x["key"], k, e)
    assert abs(x["loss"] - y["loss"]) <= 1e-5 * max(1.0, abs(y["loss"])), x["key"]
print("%d cases agree: masks equal, losses within 1e-5, per-point terms and gradients within 1e-4 of their scale (worst %.2e)" % (len(a), worst))
