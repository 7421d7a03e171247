#!/usr/bin/env python3
"""Fast / slow state (DESIGN s5): with the batch's buffers FIXED, does re-allocating the library's own LL
scratch (dwt_util_finish + init, behind spacers) move the level-0 rate?"""
import os, sys, time
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, J, nb = 8192, 5, 64
img = n * n * 4
dwt.dwt_util_init(); dwt.use_torch_stream()
def l0(src, dst):
    for _ in range(2): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n*4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(2)
    for _ in range(5): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n*4, n, n, J)
    torch.cuda.synchronize(); ms, cnt = dwt.prof_read_levels(J); dwt.prof_enable(0)
    return [2*4*(n >> j)**2*nb/ms[j]/1e6 for j in range(J)]
for batch_spacer in (0, 1.3):
    sp0 = torch.empty(int(batch_spacer * (1 << 30)), dtype=torch.uint8, device="cuda") if batch_spacer else None
    src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)
    for scratch_spacer in (0, 0.7, 3, 9, 0):
        dwt.dwt_util_finish()
        sp = torch.empty(int(scratch_spacer * (1 << 30)), dtype=torch.uint8, device="cuda") if scratch_spacer else None
        dwt.dwt_util_init(); dwt.use_torch_stream()
        r = l0(src, dst)
        print(f"batch behind {batch_spacer} GiB, scratch behind {scratch_spacer} GiB: " + " ".join(f"L{j} {x:5.0f}" for j, x in enumerate(r)) + " GB/s", flush=True)
        del sp
    del src, dst, sp0; torch.cuda.empty_cache()
