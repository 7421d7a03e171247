#!/usr/bin/env python3
"""Host-pointer (drop-in) call cost vs row pitch: dense, libdwt's prime 'optimal' stride."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import libdwt_amd as dwt
dwt.dwt_util_init()
for (w, h, J) in [(512, 512, -1), (1920, 1080, 1), (4096, 4096, 3), (8192, 8192, 5)]:
    for pitch in (w * 4, dwt.lib.dwt_util_get_opt_stride(w * 4)):
        buf = np.zeros(pitch * h + 64, np.uint8)
        img = np.ndarray((h, w), np.float32, buf, 0, (pitch, 4)) if pitch % 4 == 0 else None
        rng = np.random.default_rng(0)
        a = rng.random((h, w), dtype=np.float32)
        for y in range(h):
            buf[y * pitch:y * pitch + w * 4] = a[y].view(np.uint8)
        ts = []
        for _ in range(6):
            t0 = time.perf_counter()
            dwt.dwt_cdf97_2f_s(buf, pitch, 4, w, h, w, h, J)
            ts.append(time.perf_counter() - t0)
        print(f"{w}x{h} J={J} pitch {pitch:6d} B: median {statistics.median(ts[1:])*1e3:8.3f} ms  min {min(ts)*1e3:8.3f} ms", flush=True)
