#!/usr/bin/env python3
"""Runs the 16-image forward transform for a few seconds and prints the rate per half second;
meant to run beside a `rocm-smi` sampling loop (scripts/r02_clocks.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
nb, n, J = 16, 8192, 5
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
dwt.dwt_util_init(); dwt.use_torch_stream()
x = torch.rand((nb, n, n), device="cuda"); y = torch.empty_like(x)
torch.cuda.synchronize()
t_end = time.time() + secs
print(f"start {time.time():.2f}", flush=True)
while time.time() < t_end:
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        dwt.transform2d_batch("cdf97_s", 0, x, y, n * n * 4, nb, n * 4, n, n, J)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / 50
    print(f"{time.time():.2f}  {t:6.3f} ms  {nb*n*n/t/1e6:6.1f} Gsamples/s", flush=True)
