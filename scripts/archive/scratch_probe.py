#!/usr/bin/env python3
"""Does the rate depend on where the library's own scratch (the LL ping-pong buffers) lies?  The
scratch is released and re-allocated between timings, with dummy allocations of odd sizes in
between so that it lands somewhere else each time.  python scripts/scratch_probe.py"""
import os, sys, statistics, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
nb, n, J = 16, 8192, 5
dwt.dwt_util_init(); dwt.use_torch_stream()
x = torch.rand((nb, n, n), device="cuda"); y = torch.empty_like(x)
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
keep = []
def timed():
    for _ in range(3):
        dwt.transform2d_batch("cdf97_s", 0, x, y, n * n * 4, nb, n * 4, n, n, J)
    torch.cuda.synchronize(); ts = []
    for _ in range(8):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); dwt.transform2d_batch("cdf97_s", 0, x, y, n * n * 4, nb, n * 4, n, n, J); e1.record()
        e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)
for k in range(10):
    t = timed()
    print(f"placement {k}: {t:6.3f} ms  {nb*n*n/t/1e6:6.1f} Gsamples/s", flush=True)
    torch.cuda.synchronize()
    dwt.dwt_util_finish()                       # releases the scratch
    sz = random.randrange(64, 1600) * (1 << 20) + random.randrange(0, 512) * 4096
    keep.append(torch.empty(sz, dtype=torch.uint8, device="cuda"))   # takes the hole (or part of it)
    dwt.use_torch_stream()
