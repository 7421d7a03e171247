#!/usr/bin/env python3
"""Per-level kernel times of the batched forward 9/7 (8192^2, J=5) for values of one option, alternated in one
process (HIP events per level): python scripts/archive/r04/levels_ab.py name=v0,v1,... [images]"""
import os, sys, statistics
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
name, vals = sys.argv[1].split("=")
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n, J = 8192, 5
dwt.dwt_util_init(); dwt.use_torch_stream()
src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)
for rnd in range(2):
    for v in vals.split(","):
        dwt.set_option(name, int(v))
        for _ in range(3): dwt.transform2d_batch("cdf97_s", 0, src, dst, n*n*4, nb, n*4, n, n, J)
        torch.cuda.synchronize(); dwt.prof_enable(2)
        for _ in range(8): dwt.transform2d_batch("cdf97_s", 0, src, dst, n*n*4, nb, n*4, n, n, J)
        torch.cuda.synchronize(); ms, cnt = dwt.prof_read_levels(J); dwt.prof_enable(False)
        print(f"{name}={v}: " + "  ".join(f"L{j} {ms[j]*1e3:8.1f} us" for j in range(J)) + f"  sum {sum(ms)*1e3:8.1f} us", flush=True)
