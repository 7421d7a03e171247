#!/usr/bin/env python3
"""Out-of-place 3-D forward (config 5 entry): fused one-pass levels vs the two-pass path.
python scripts/archive/vol_op_bench.py [n] [levels]"""
import os, sys, time, statistics
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lv = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dwt.dwt_util_init(); dwt.use_torch_stream()
a = torch.rand((n, n, n), device="cuda"); b = torch.empty_like(a)
variants = os.environ.get("VARIANTS", "vol_fused=1;vol_fused=0;vol_fused=1,vol_nt=0;vol_fused=1,vol_nt=2;vol_fused=1,vol_tile_pairs=64;vol_fused=1,vol_tile_pairs=256;vol_fused=1").split(";")
D = dict(vol_fused=1, vol_nt=-1, vol_tile_pairs=0, vol_swizzle=1, vol_rows=8, vol_direct=2, vol_whole=1, vol_ip_waves=0)
for opts in variants:
    for k, v in D.items(): dwt.set_option(k, v)
    for kv in [x for x in opts.split(",") if x]:
        k, v = kv.split("="); dwt.set_option(k, int(v))
    for levels in (1, lv):
        fn = lambda: dwt.transform3d_op(a, b, n * 4, n * n * 4, n, n, n, levels)
        for _ in range(2): fn()
        torch.cuda.synchronize(); ts = []
        for _ in range(7):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = statistics.median(ts)
        print(f"{opts:34s} {n}^3 {levels} level(s): {t*1e3:8.3f} ms  {n**3/t/1e9:7.1f} Gvoxel/s", flush=True)
