#!/usr/bin/env python3
"""Per-image time of the batched forward call (8192^2, J = 5) by images per call: does a chunk small enough for its LL
bands to live in the 256 MiB Infinity Cache beat the big launches?  64 images resident, processed in chunks of NB."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, J, total = 8192, 5, 64
src = torch.rand((total, n, n), device="cuda"); dst = torch.empty_like(src)
for nb in (64, 32, 16, 8, 4, 2, 1):
    def step():
        for k in range(0, total, nb):
            dwt.transform2d_batch("cdf97_s", 0, src[k:k + nb], dst[k:k + nb], n * n * 4, nb, n * 4, n, n, J)
    for _ in range(2): step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    t = statistics.median(a.elapsed_time(b) for a, b in ev)
    print(f"chunks of {nb:2d}: {t:7.3f} ms per 64 images = {t / total * 1e3:6.1f} us/image = {total * n * n / t / 1e6:6.1f} Gsamples/s", flush=True)
