#!/usr/bin/env python3
"""Fast / slow state (DESIGN s5): does the rate depend on the distance between the source and the
destination batch?  ONE allocation holds both; the destination starts `delta` bytes behind the end of
the source.  Several processes in a row (the state changes from process to process)."""
import os, sys, time
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, J, nb = 8192, 5, int(os.environ.get("IMAGES", 64))
dwt.dwt_util_init(); dwt.use_torch_stream()
img = n * n * 4
deltas = [0, 4096, 1 << 20, 3 << 20, 32 << 20, 100 << 20, 1 << 30, (1 << 30) + (3 << 20), 0]
slab = torch.empty(2 * nb * img + max(deltas) + (1 << 21), dtype=torch.uint8, device="cuda")
src = slab[:nb * img].view(torch.float32).view(nb, n, n)
src.uniform_()
for d in deltas:
    dst = slab[nb * img + d: nb * img + d + nb * img].view(torch.float32).view(nb, n, n)
    for _ in range(3): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n*4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(10): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n*4, n, n, J)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10
    ms, cnt = dwt.prof_read(); dwt.prof_enable(False)
    print(f"delta {d:>12d} B: {nb*n*n/el/1e9:7.1f} Gsamples/s  L0 {2*4*n*n*nb/(ms/cnt)/1e6:7.0f} GB/s", flush=True)
