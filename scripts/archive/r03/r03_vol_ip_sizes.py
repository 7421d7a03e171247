#!/usr/bin/env python3
"""In-place 3-D level, one pass over a halo snapshot against two passes through scratch, per size:
python scripts/archive/r03/r03_vol_ip_sizes.py [n ...]   (vol_fused=2 forces the one-pass level where the default would not take it)"""
import os, sys, time, statistics
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
sizes = [int(x) for x in sys.argv[1:]] or [256, 384, 448, 512, 640, 768]
for n in sizes:
    a = torch.rand((n, n, n), device="cuda")
    res = {}
    for rnd in range(2):
        for name, opts in (("one pass", {"vol_fused": 2, "vol_inplace_fused": 1}), ("two passes", {"vol_fused": 1, "vol_inplace_fused": 0})):
            for k, v in opts.items():
                dwt.set_option(k, v)
            for inverse in (0, 1):
                fn = lambda: dwt.transform3d(inverse, a, n * 4, n * n * 4, n, n, n, 1)
                fn(); fn(); torch.cuda.synchronize(); ts = []
                for _ in range(9):
                    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                res.setdefault((name, inverse), []).append(statistics.median(ts))
    dwt.set_option("vol_fused", 1); dwt.set_option("vol_inplace_fused", 1)
    line = f"{n}^3: "
    for inverse in (0, 1):
        o = min(res[("one pass", inverse)]); t = min(res[("two passes", inverse)])
        line += f"{'inverse' if inverse else 'forward'} one pass {o*1e3:7.3f} ms / two passes {t*1e3:7.3f} ms   "
    print(line, flush=True)
    del a
