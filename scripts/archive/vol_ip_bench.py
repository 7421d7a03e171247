#!/usr/bin/env python3
"""In-place 3-D transform (dwt_hip_transform3d), forward and inverse, 1 and 3 levels.
python scripts/archive/vol_ip_bench.py [n]"""
import os, sys, time, statistics
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dwt.dwt_util_init(); dwt.use_torch_stream()
a = torch.rand((n, n, n), device="cuda")
for levels in (1, 3):
    for inverse in (0, 1):
        fn = lambda: dwt.transform3d(inverse, a, n * 4, n * n * 4, n, n, n, levels)
        for _ in range(2): fn()
        torch.cuda.synchronize(); ts = []
        for _ in range(7):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = statistics.median(ts)
        print(f"in place {'inverse' if inverse else 'forward'} {n}^3 {levels} level(s): {t*1e3:8.3f} ms  {n**3/t/1e9:7.1f} Gvoxel/s", flush=True)
