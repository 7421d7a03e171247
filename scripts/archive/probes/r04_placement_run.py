#!/usr/bin/env python3
"""Workload of scripts/archive/r04/placement_pmc.sh: several physical placements of the batch inside one
process, a few 2-level forward transforms of 64 x 8192^2 on each (run under rocprofv3)."""
import os, sys
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, nb, J = 8192, 64, 2
img = n * n * 4
dwt.dwt_util_init(); dwt.use_torch_stream()
hold = []
for k, sp in enumerate([0, 1.3, 7, 2.6, 11, 23]):
    spacer = torch.empty(int(sp * (1 << 30)), dtype=torch.uint8, device="cuda") if sp else None
    src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)
    for _ in range(4):
        dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n * 4, n, n, J)
    torch.cuda.synchronize()
    hold.append(spacer)
    del src, dst; torch.cuda.empty_cache()
