#!/usr/bin/env python3
"""Fast / slow state (DESIGN s5): is it the PLATFORM's?  For a series of placements of the batch (two
allocations of 16 GiB behind spacers of various sizes) print the level-0 kernel's rate next to the rate
of a plain device copy between the same two buffers and of a read-only / write-only pass over each."""
import os, sys, time
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, J, nb = 8192, 5, 64
dwt.dwt_util_init(); dwt.use_torch_stream()
img = n * n * 4
def ev(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
for spacer in (0, 1.3, 0, 7, 2.6, 0, 11, 1.3):
    sp = torch.empty(int(spacer * (1 << 30)), dtype=torch.uint8, device="cuda") if spacer else None
    src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)
    for _ in range(2): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n*4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(True)
    for _ in range(6): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n*4, n, n, J)
    torch.cuda.synchronize(); ms, cnt = dwt.prof_read(); dwt.prof_enable(False)
    l0 = 2 * 4 * n * n * nb / (ms / cnt) / 1e6
    tc = ev(lambda: dst.copy_(src))
    tr = ev(lambda: src.sum())
    tw = ev(lambda: dst.fill_(1.0))
    tr2 = ev(lambda: dst.sum())
    print(f"spacer {spacer:5.1f} GiB: level 0 {l0:6.0f} GB/s | copy {2*nb*img/tc/1e9:6.0f} GB/s | read src {nb*img/tr/1e9:6.0f}  read dst {nb*img/tr2/1e9:6.0f}  fill dst {nb*img/tw/1e9:6.0f} GB/s", flush=True)
    del src, dst, sp; torch.cuda.empty_cache()
