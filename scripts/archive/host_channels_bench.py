#!/usr/bin/env python3
"""Host-pointer call on ONE channel of an interleaved multi-channel image (the calling convention of
the reference's OpenCV wrapper, src/cvdwt.cpp:98-135: stride_y = channels * sizeof(T))."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import libdwt_amd as dwt
dwt.dwt_util_init()
for (w, h, c, J) in [(512, 384, 3, 4), (1920, 1080, 3, 4), (4096, 4096, 3, 5), (4096, 4096, 1, 5), (8192, 8192, 3, 5)]:
    img = np.random.default_rng(0).random((h, w, c), dtype=np.float32)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        dwt.dwt_cdf97_2f_s(img.ctypes.data + 4, img.strides[0], img.strides[1], w, h, w, h, J) if c > 1 else \
            dwt.dwt_cdf97_2f_s(img, img.strides[0], 4, w, h, w, h, J)
        ts.append(time.perf_counter() - t0)
    print(f"{w}x{h} x{c} channels, J={J}: one channel forward  median {statistics.median(ts[1:])*1e3:8.3f} ms  min {min(ts)*1e3:8.3f} ms  "
          f"{w*h/min(ts)/1e9:6.2f} Gsamples/s", flush=True)
