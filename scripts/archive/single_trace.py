#!/usr/bin/env python3
"""Workload for a kernel-trace of the single-image entries: 30 out-of-place calls, then 30 in-place
calls of dwt_cdf97_2f_s on rotating 8192^2 images.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02/trace_single -- python3 scripts/single_trace.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt

n, J, nb = 8192, 5, 4
dwt.dwt_util_init(); dwt.use_torch_stream()
for kv in sys.argv[1:]:
    k, v = kv.split("="); dwt.set_option(k, int(v))
src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)
for i in range(30):
    dwt.dwt_cdf97_2f_s2(src[i % nb], dst[i % nb], n * 4, 4, n, n, n, n, J)
torch.cuda.synchronize()
dst.copy_(src)
torch.cuda.synchronize()
for i in range(30):
    dwt.dwt_cdf97_2f_s(dst[i % nb], n * 4, 4, n, n, n, n, J)
torch.cuda.synchronize()
print("done")
