#!/usr/bin/env python3
"""Does it pay to have the detail bands of the deeper levels in the cache before an inverse call of one 8192^2 image reads
them?  Per-level kernel times of dwt_cdf97_2i_s2 (rotating over 8 images, so every call starts cold) with and without a
read of the image's top-left 4096 x 4096 corner (levels >= 1) just before the call."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, J, nb = 8192, 5, 8
img = torch.rand((nb, n, n), device="cuda"); co = torch.empty_like(img); out = torch.empty_like(img)
dwt.transform2d_batch("cdf97_s", 0, img, co, n * n * 4, nb, n * 4, n, n, J)
sink = torch.zeros(1, device="cuda")
for corner in (0, 4096, 2048, 0, 4096, 2048):
    dwt.prof_enable(2)
    for i in range(48):
        k = i % nb
        if corner:
            sink += co[k, :corner, :corner].sum()
        dwt.dwt_cdf97_2i_s2(co[k], out[k], n * 4, 4, n, n, n, n, J)
    torch.cuda.synchronize()
    ms, cnt = dwt.prof_read_levels(J); dwt.prof_enable(0)
    print(f"pre-read corner {corner:4d}: levels (us) " + " ".join(f"L{j}:{ms[j] * 1e3:.1f}" for j in range(J)) + f"  sum {sum(ms[:J]) * 1e3:.1f}", flush=True)
