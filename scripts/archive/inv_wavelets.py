#!/usr/bin/env python3
"""Level-0 kernel time of the single-image inverse (and forward) for the float 9/7 and 5/3 wavelets:
is the inverse bound by its arithmetic or by its memory pattern?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, J, nb = 8192, 5, 8
dwt.dwt_util_init(); dwt.use_torch_stream()
x = torch.rand((nb, n, n), device="cuda"); y = torch.empty_like(x)
for wav in ("cdf97_s", "cdf53_s"):
    for inv in (0, 1):
        fn = lambda i: dwt.transform2d_batch(wav, inv, x[i % nb], y[i % nb], n * n * 4, 1, n * 4, n, n, J)
        for i in range(8): fn(i)
        dwt.prof_enable(2)
        for i in range(24): fn(i)
        torch.cuda.synchronize()
        ms, cnt = dwt.prof_read_levels(J); dwt.prof_enable(0)
        print(f"{wav} {'inverse' if inv else 'forward'}: " + " ".join(f"L{j}:{ms[j]*1e3:6.1f}" for j in range(J)), flush=True)
