#!/usr/bin/env python3
"""One channel of a 1920 x 1080 x 3 float matrix through dwt_cdf97_2f_s / _2i_s (4 levels), the way
src/cvdwt.cpp:98-135 calls the entries: the matrix in host memory (CPU repack + PCIe) against the matrix
resident in HBM (pack / unpack kernels); also the dense single-channel device image for scale."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libdwt_amd as dwt  # noqa: E402

dwt.dwt_util_init()
h, w, c = 1080, 1920, 3
img = np.random.default_rng(0).random((h, w, c), dtype=np.float32)
out = {}


def timeit(f, n=30):
    for _ in range(5):
        f()
    dwt.sync()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        dwt.sync()
        ts.append((time.perf_counter() - t0) * 1e6)
    return {"median_us": round(float(np.median(ts)), 1), "min_us": round(float(min(ts)), 1)}


host = img.copy()
out["host_fwd"] = timeit(lambda: dwt.dwt_cdf97_2f_s(host.ctypes.data + 4, host.strides[0], host.strides[1], w, h, w, h, 4))
out["host_inv"] = timeit(lambda: dwt.dwt_cdf97_2i_s(host.ctypes.data + 4, host.strides[0], host.strides[1], w, h, w, h, 4))
d = dwt.lib.dwt_hip_malloc(img.nbytes)
dwt.lib.dwt_hip_memcpy_h2d(d, img.ctypes.data, img.nbytes)
out["device_fwd"] = timeit(lambda: dwt.dwt_cdf97_2f_s(d + 4, img.strides[0], img.strides[1], w, h, w, h, 4))
out["device_inv"] = timeit(lambda: dwt.dwt_cdf97_2i_s(d + 4, img.strides[0], img.strides[1], w, h, w, h, 4))
dense = dwt.DeviceImage(h, w).upload(np.ascontiguousarray(img[:, :, 1]))
out["dense_device_fwd"] = timeit(lambda: dwt.dwt_cdf97_2f_s(dense.ptr, dense.stride_x, 4, w, h, w, h, 4))
out["dense_device_inv"] = timeit(lambda: dwt.dwt_cdf97_2i_s(dense.ptr, dense.stride_x, 4, w, h, w, h, 4))
# kernel time of the strided call, per kernel class
dwt.prof_enable(True)
for _ in range(20):
    dwt.dwt_cdf97_2f_s(d + 4, img.strides[0], img.strides[1], w, h, w, h, 4)
ms, n = dwt.prof_read()
out["device_fwd_levels_ms_per_call"] = round(ms / 20, 4)
dwt.prof_enable(False)
print(json.dumps(out))
