#!/usr/bin/env python3
"""Out-of-place 3-D forward, 1024^3, 3 levels, on volumes placed by dwt_hip_alloc_volumes: A/B of an option
(HIP events, median of 20): python scripts/archive/r04/vol_ab.py name=v0,v1"""
import os, sys, statistics
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
name, vals = sys.argv[1].split("=")
n, J = 1024, 3
dwt.dwt_util_init(); dwt.use_torch_stream()
src, dst = dwt.alloc_volumes(n, n, n, J)
dwt.lib.dwt_hip_probe_pair_us(src, None, n ** 3 * 4)
fn = lambda: dwt.transform3d_op(src, dst, n * 4, n * n * 4, n, n, n, J)
for rnd in range(3):
    for v in vals.split(","):
        dwt.set_option(name, int(v))
        for _ in range(5): fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        print(f"{name}={v}: {statistics.median(a.elapsed_time(b) for a, b in evs):.4f} ms", flush=True)
