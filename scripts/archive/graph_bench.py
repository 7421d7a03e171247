#!/usr/bin/env python3
"""Single 8192^2 5-level forward transform: direct launches against HIP graph replay
(out of place and in place).  python scripts/graph_bench.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init()
n, J = 8192, 5
x = torch.rand((n, n), device="cuda"); y = torch.empty_like(x)
s = torch.cuda.Stream()

def timed(fn, reps=100, warm=20):
    for _ in range(warm): fn()
    s.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(s); fn(); e1.record(s); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts), min(ts)

with torch.cuda.stream(s):
    dwt.use_torch_stream()
    for name, call in (("out of place", lambda: dwt.dwt_cdf97_2f_s2(x, y, n * 4, 4, n, n, n, n, J)),
                       ("in place", lambda: dwt.dwt_cdf97_2f_s(x, n * 4, 4, n, n, n, n, J))):
        call(); s.synchronize()
        med, mn = timed(call)
        print(f"{name:14s} direct launches: median {med:7.1f} us  min {mn:7.1f} us", flush=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            dwt.use_torch_stream()
            call()
        med, mn = timed(g.replay)
        print(f"{name:14s} graph replay   : median {med:7.1f} us  min {mn:7.1f} us", flush=True)
