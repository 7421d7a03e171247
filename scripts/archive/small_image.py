#!/usr/bin/env python3
"""Device-resident small images (BASELINE config 1: 512^2, full depth as examples/simple runs it):
HIP-event time per call and the kernels of one call."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
for n, J in ((512, -1), (512, 1), (256, -1), (1024, -1), (2048, -1)):
    x = torch.rand((n, n), device="cuda"); y = torch.empty_like(x)
    for name, fn in (("forward s2", lambda: dwt.dwt_cdf97_2f_s2(x, y, n * 4, 4, n, n, n, n, J)),
                     ("forward in place", lambda: dwt.dwt_cdf97_2f_s(y, n * 4, 4, n, n, n, n, J)),
                     ("inverse s2", lambda: dwt.dwt_cdf97_2i_s2(y, x, n * 4, 4, n, n, n, n, J))):
        for _ in range(10): fn()
        torch.cuda.synchronize(); ts = []
        for _ in range(50):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"{n}^2 J={J:2d} {name:18s}: median {statistics.median(ts):7.1f} us  min {min(ts):7.1f} us", flush=True)
