"""outputs of tools/device_check/float_check.cpp on the device and under host emulation, tensor by tensor:
    python tools/device_check/float_compare.py gpurun_out/float_check_dev.bin /tmp/float_check_host.bin"""
import sys

import numpy as np


def read(path):
    raw = open(path, "rb").read()
    out, o = [], 0
    while o < len(raw):
        name = raw[o:o + 32].split(b"\0")[0].decode(); o += 32
        n = int(np.frombuffer(raw, np.uint32, 1, o)[0]); o += 4
        out.append((name, np.frombuffer(raw, np.float32, n, o))); o += 4 * n
    return out


a, b = read(sys.argv[1]), read(sys.argv[2])
assert [x[0] for x in a] == [x[0] for x in b]
for (name, x), (_, y) in zip(a, b):
    same = np.array_equal(x.view(np.uint32), y.view(np.uint32))
    scale = max(float(np.abs(y).max()), 1e-30)
    err = float(np.abs(x - y).max()) / scale
    print("%-26s %8d floats  %s  max |device - emulator| / max|x| = %.2e" % (name, len(x), "BYTE-IDENTICAL" if same else "              ", err))
    assert np.isfinite(x).all() and err < 1e-4, name
