#!/usr/bin/env python3
"""Whole calls (J = 5, 8192^2) with the library given by DWT_HIP_LIB: forward / inverse `_s2` on one image (rotating over
8), in-place forward, batches of 8 forward / inverse, the interleaved forward.  HIP-event medians.  Run once per library
build, alternated by the shell, on ONE box."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, J, nb = 8192, 5, 8
img = torch.rand((nb, n, n), device="cuda"); co = torch.empty_like(img); out = torch.empty_like(img)
dwt.transform2d_batch("cdf97_s", 0, img, co, n * n * 4, nb, n * 4, n, n, J)
work = img.clone()
def timed(fn, reps):
    for i in range(4): fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(i); b.record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) * 1e3 for a, b in ev]
    return statistics.median(t), min(t)
legs = [("2f_s2 x1", lambda i: dwt.dwt_cdf97_2f_s2(img[i % nb], out[i % nb], n * 4, 4, n, n, n, n, J), 60),
        ("2i_s2 x1", lambda i: dwt.dwt_cdf97_2i_s2(co[i % nb], out[i % nb], n * 4, 4, n, n, n, n, J), 60),
        ("2f_s  x1", lambda i: dwt.dwt_cdf97_2f_s(work[i % nb], n * 4, 4, n, n, n, n, J), 60),
        ("fwd   x8", lambda i: dwt.transform2d_batch("cdf97_s", 0, img, out, n * n * 4, nb, n * 4, n, n, J), 12),
        ("inv   x8", lambda i: dwt.transform2d_batch("cdf97_s", 1, co, out, n * n * 4, nb, n * 4, n, n, J), 12),
        ("il fwd x1", lambda i: dwt.transform2d_interleaved("cdf97_s", 0, 0, img[i % nb], out[i % nb], n * 4, 4, n, n, n, n, J), 40),
        ("53i fwd x8", None, 12)]
i32 = torch.randint(-32768, 32768, (nb, 4096, 4096), device="cuda", dtype=torch.int32); o32 = torch.empty_like(i32)
legs[-1] = ("53i fwd x8", lambda i: dwt.transform2d_batch("cdf53_i", 0, i32, o32, 4096 * 4096 * 4, nb, 4096 * 4, 4096, 4096, 3), 12)
print(os.environ.get("DWT_HIP_LIB", "(shipped build)"))
for name, fn, reps in legs:
    med, mn = timed(fn, reps)
    print(f"    {name:10s} median {med:8.1f} us  min {mn:8.1f}", flush=True)

# per-level kernel times of the two `_s2` calls (HIP events around every level: they add a little to each)
for name, fn in (("2f_s2", legs[0][1]), ("2i_s2", legs[1][1])):
    dwt.prof_enable(2)
    for i in range(40): fn(i)
    torch.cuda.synchronize()
    ms, cnt = dwt.prof_read_levels(J); dwt.prof_enable(0)
    print(f"    {name} levels: " + " ".join(f"L{j}:{ms[j] * 1e3:.1f}" for j in range(J)), flush=True)
