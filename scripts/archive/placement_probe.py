#!/usr/bin/env python3
"""Does the rate of the same batch transform depend on WHERE its buffers were allocated?
Several (src, dst) sets are allocated one after the other in one process (all kept alive) and the
same 16-image forward transform is timed on each.  python scripts/placement_probe.py [sets] [images]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n, J = 8192, 5
dwt.dwt_util_init(); dwt.use_torch_stream()
sets = []
for k in range(nsets):
    x = torch.rand((nb, n, n), device="cuda"); y = torch.empty_like(x)
    sets.append((x, y))
def run(x, y):
    dwt.transform2d_batch("cdf97_s", 0, x, y, n * n * 4, nb, n * 4, n, n, J)
def timed(x, y, reps=8):
    for _ in range(3): run(x, y)
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); run(x, y); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)
for rnd in range(2):
    for k, (x, y) in enumerate(sets):
        t = timed(x, y)
        print(f"round {rnd} set {k}: src {x.data_ptr():#x} dst {y.data_ptr():#x}  {t:7.3f} ms  {nb*n*n/t/1e6:6.1f} Gsamples/s", flush=True)
# cross pairs: is it the source or the destination?
x0, y0 = sets[0]; x1, y1 = sets[1]
for name, (x, y) in (("src0->dst1", (x0, y1)), ("src1->dst0", (x1, y0))):
    t = timed(x, y)
    print(f"{name}: {t:7.3f} ms  {nb*n*n/t/1e6:6.1f} Gsamples/s", flush=True)
