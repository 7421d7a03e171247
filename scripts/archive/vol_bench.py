#!/usr/bin/env python3
"""3-D path timing (config 5): python scripts/vol_bench.py [n] [levels]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lv = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dwt.dwt_util_init(); dwt.use_torch_stream()
V = torch.rand((n, n, n), device="cuda")
for _ in range(2): dwt.transform3d(0, V, n*4, n*n*4, n, n, n, lv)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): dwt.transform3d(0, V, n*4, n*n*4, n, n, n, lv)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / 5
print(f"{n}^3 {lv} levels: {el*1e3:.3f} ms  {n**3/el/1e9:.1f} Gvoxel/s")
