#!/usr/bin/env python3
"""In-place 3-D level: tiles of 32 rows (4 waves, two workgroups per CU) against tiles of 64 rows (8 waves, one),
forward and inverse, per size: python scripts/archive/r03/r03_vol_ip_waves.py [n ...]"""
import os, sys, time, statistics
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
sizes = [int(x) for x in sys.argv[1:]] or [512, 576, 640, 704, 768, 832, 896, 960, 1024, 1152]
dwt.set_option("vol_fused", 2)
for n in sizes:
    a = torch.rand((n, n, n), device="cuda")
    res = {}
    for rnd in range(2):
        for w in (4, 8):
            dwt.set_option("vol_ip_waves", w)
            for inverse in (0, 1):
                fn = lambda: dwt.transform3d(inverse, a, n * 4, n * n * 4, n, n, n, 1)
                fn(); fn(); torch.cuda.synchronize(); ts = []
                for _ in range(7):
                    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                res.setdefault((w, inverse), []).append(statistics.median(ts))
    print(f"{n}^3: " + "  ".join(f"{'inv' if i else 'fwd'} 4w {min(res[(4,i)])*1e3:6.3f} 8w {min(res[(8,i)])*1e3:6.3f} ms" for i in (0, 1)), flush=True)
    del a
dwt.set_option("vol_ip_waves", 4); dwt.set_option("vol_fused", 1)
