#!/usr/bin/env python3
"""Does the small two-stream write probe (dwt_hip_probe_pair_us) see the same coarse regions as the
real level-0 kernel?  One pool; per GiB: probe(ref at 0, chunk at P) beside the level-0 rate with the
destination at 2 and the LL scratch at P."""
import os, sys
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
GB = 1 << 30
dwt.dwt_util_init(); dwt.use_torch_stream()
free_b, total_b = torch.cuda.mem_get_info()
pool_gb = (free_b - 4 * GB) // GB
pool = torch.empty(pool_gb * GB, dtype=torch.uint8, device="cuda")
base = pool.data_ptr()
PB = int(os.environ.get("PROBE_MB", 256)) << 20
for ref in (0, 100, 70):
    row = []
    for P in range(0, pool_gb - 1):
        if P == ref: row.append("---"); continue
        us = dwt.lib.dwt_hip_probe_pair_us(base + ref * GB, base + P * GB, PB)
        row.append(f"{2 * PB / us / 1e3:.0f}")
    print(f"-- probe pair rate GB/s (2 x {PB >> 20} MiB), reference at {ref} GiB, other at P = 0, 1, ...\n" + " ".join(row), flush=True)
row = [f"{PB / dwt.lib.dwt_hip_probe_pair_us(base + P * GB, None, PB) / 1e3:.0f}" for P in range(0, pool_gb - 1, 4)]
print("-- single stream GB/s at P = 0, 4, ...\n" + " ".join(row), flush=True)
