#!/usr/bin/env python3
"""Does the fast / slow state of a process (DESIGN s5) depend on WHERE its buffers lie?  One process:
allocate the bench's batch, measure, free everything (empty_cache), allocate again -- optionally behind
a spacer allocation that shifts the placement -- and measure again."""
import os, sys, time
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, J, nb = 8192, 5, int(os.environ.get("IMAGES", 64))
dwt.dwt_util_init(); dwt.use_torch_stream()
def measure(tag, spacer_gb=0):
    sp = torch.empty(int(spacer_gb * (1 << 30)), dtype=torch.uint8, device="cuda") if spacer_gb else None
    src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)
    for _ in range(3): dwt.transform2d_batch("cdf97_s", 0, src, dst, n*n*4, nb, n*4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(10): dwt.transform2d_batch("cdf97_s", 0, src, dst, n*n*4, nb, n*4, n, n, J)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10
    ms, cnt = dwt.prof_read(); dwt.prof_enable(False)
    print(f"{tag:40s} {nb*n*n/el/1e9:7.1f} Gsamples/s  L0 {2*4*n*n*nb/(ms/cnt)/1e6:7.0f} GB/s  src@{src.data_ptr():#x} dst@{dst.data_ptr():#x}", flush=True)
    del src, dst, sp
    torch.cuda.empty_cache()
measure("first allocation")
measure("after free + empty_cache")
measure("behind a 3 GB spacer", 3)
measure("behind a 20 GB spacer", 20)
measure("behind a 1.3 GB spacer", 1.3)
measure("plain again")
measure("behind a 50 GB spacer", 50)
measure("plain again")
