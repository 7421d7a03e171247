#!/usr/bin/env python3
"""Workload for a kernel trace of the interleaved entries (8192^2 J=5, out of place): 10 forward, 10 inverse."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, J = 8192, 5
dwt.dwt_util_init(); dwt.use_torch_stream()
a = torch.rand((n, n), device="cuda"); b = torch.empty_like(a); c = torch.empty_like(a)
for _ in range(10):
    dwt.transform2d_interleaved("cdf97_s", 0, 0, a, b, n*4, 4, n, n, None, None, J)
torch.cuda.synchronize()
for _ in range(10):
    dwt.transform2d_interleaved("cdf97_s", 1, 0, b, c, n*4, 4, n, n, None, None, J)
torch.cuda.synchronize()
print("done")
