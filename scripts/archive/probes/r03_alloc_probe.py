#!/usr/bin/env python3
"""Fast / slow state (DESIGN s5) against HOW the batch's two buffers are allocated, alternated in one process."""
import os, sys, time, ctypes
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, J, nb = 8192, 5, int(os.environ.get("IMAGES", 64))
dwt.dwt_util_init(); dwt.use_torch_stream()
img = n * n * 4
def run(tag, src, dst):
    src.uniform_()
    for _ in range(3): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n*4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(10): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n*4, n, n, J)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10
    ms, cnt = dwt.prof_read(); dwt.prof_enable(False)
    print(f"{tag:46s} {nb*n*n/el/1e9:7.1f} Gsamples/s  L0 {2*4*n*n*nb/(ms/cnt)/1e6:7.0f} GB/s", flush=True)
def view(t, off): return t[off:off + nb * img].view(torch.float32).view(nb, n, n)
for rnd in range(2):
    a = torch.empty(nb * img, dtype=torch.uint8, device="cuda"); b = torch.empty(nb * img, dtype=torch.uint8, device="cuda")
    run("two allocations of 16 GiB", view(a, 0), view(b, 0)); del a, b; torch.cuda.empty_cache()
    s = torch.empty(2 * nb * img, dtype=torch.uint8, device="cuda")
    run("one allocation of 32 GiB", view(s, 0), view(s, nb * img)); del s; torch.cuda.empty_cache()
    a = torch.empty(nb * img + (64 << 20), dtype=torch.uint8, device="cuda"); b = torch.empty(nb * img + (64 << 20), dtype=torch.uint8, device="cuda")
    run("two allocations of 16 GiB + 64 MiB", view(a, 0), view(b, 0)); del a, b; torch.cuda.empty_cache()
    parts = [torch.empty(nb * img // 4, dtype=torch.uint8, device="cuda") for _ in range(8)]
    # (views must be contiguous: use the first two quarters-of-quarters only when they happen to be adjacent -- skip)
    del parts; torch.cuda.empty_cache()
    a = torch.empty(nb * img, dtype=torch.uint8, device="cuda"); pad = torch.empty(5 << 30, dtype=torch.uint8, device="cuda"); b = torch.empty(nb * img, dtype=torch.uint8, device="cuda")
    run("16 GiB, 5 GiB in between, 16 GiB", view(a, 0), view(b, 0)); del a, b, pad; torch.cuda.empty_cache()
    s = torch.empty(2 * nb * img + (1 << 30), dtype=torch.uint8, device="cuda")
    run("one allocation of 33 GiB, dst first", view(s, nb * img + (1 << 30)), view(s, 0)); del s; torch.cuda.empty_cache()
