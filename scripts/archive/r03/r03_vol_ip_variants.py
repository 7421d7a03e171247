#!/usr/bin/env python3
"""In-place 3-D level (SIZE^3, default 1024; 1 level), variants alternated inside one process:
python scripts/archive/r03/r03_vol_ip_variants.py name=v1,v2 [name=...]   e.g. vol_nt=1,3 vol_tile_pairs=0,64"""
import os, sys, time, statistics, itertools
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n = int(os.environ.get("SIZE", 1024))
dwt.dwt_util_init(); dwt.use_torch_stream()
a = torch.rand((n, n, n), device="cuda")
axes = [(kv.split("=")[0], [int(x) for x in kv.split("=")[1].split(",")]) for kv in sys.argv[1:]]
combos = list(itertools.product(*[v for _, v in axes])) or [()]
res = {}
for rnd in range(3):
    for combo in combos:
        for (name, _), val in zip(axes, combo):
            dwt.set_option(name, val)
        for inverse in (0, 1):
            fn = lambda: dwt.transform3d(inverse, a, n * 4, n * n * 4, n, n, n, 1)
            fn(); torch.cuda.synchronize(); ts = []
            for _ in range(5):
                t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            res.setdefault((combo, inverse), []).append(statistics.median(ts))
for (combo, inverse), v in sorted(res.items()):
    tag = " ".join(f"{name}={val}" for (name, _), val in zip(axes, combo))
    print(f"{tag:40s} {'inverse' if inverse else 'forward'}: " + " ".join(f"{x*1e3:7.3f}" for x in v) + " ms", flush=True)
