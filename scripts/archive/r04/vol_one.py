#!/usr/bin/env python3
"""A few out-of-place 3-D forward calls (1024^3, 3 levels, placed volumes), then in-place forward / inverse ones, for a
kernel trace: rocprofv3 --kernel-trace -- python3 scripts/archive/r04/vol_one.py ; python3 scripts/archive/r04/calls_timeline.py <dir> <call>"""
import os, sys
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, J = 1024, 3
dwt.dwt_util_init(); dwt.use_torch_stream()
src, dst = dwt.alloc_volumes(n, n, n, J)
for _ in range(6):
    dwt.transform3d_op(src, dst, n * 4, n * n * 4, n, n, n, J)
    torch.cuda.synchronize()
for inv in (0, 1):
    for _ in range(4):
        dwt.transform3d(inv, dst, n * 4, n * n * 4, n, n, n, J)
        torch.cuda.synchronize()
