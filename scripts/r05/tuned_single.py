#!/usr/bin/env python3
"""One 8192^2 image through dwt_cdf97_2f_s2 / _2i_s2 with and without dwt_hip_tune (DWT_HIP_TUNE_VERBOSE=1 prints the tuner's
candidates), and the 8- and 64-image batches: what the wider candidate set of round 5 buys."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, J, nb = 8192, 5, 16
src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)

def timed(fn, reps=60):
    for i in range(8): fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(i); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[0]

fwd = lambda i: dwt.dwt_cdf97_2f_s2(src[i % nb], dst[i % nb], n * 4, 4, n, n, n, n, J)
inv = lambda i: dwt.dwt_cdf97_2i_s2(dst[i % nb], src[i % nb], n * 4, 4, n, n, n, n, J)
for rnd in range(2):
    dwt.dwt_util_finish()
    print("untuned  fwd median %.1f min %.1f us" % timed(fwd), " inv median %.1f min %.1f us" % timed(inv), flush=True)
    dwt.tune("cdf97_s", 0, src[0], dst[0], 0, 1, n * 4, n, n, J)
    dwt.tune("cdf97_s", 1, dst[0], src[0], 0, 1, n * 4, n, n, J)
    print("tuned    fwd median %.1f min %.1f us" % timed(fwd), " inv median %.1f min %.1f us" % timed(inv), flush=True)
for k in (8, 16):
    b = lambda i: dwt.transform2d_batch("cdf97_s", 0, src[:k], dst[:k], n * n * 4, k, n * 4, n, n, J)
    dwt.dwt_util_finish()
    u = timed(b, 20)
    dwt.tune("cdf97_s", 0, src[:k], dst[:k], n * n * 4, k, n * 4, n, n, J)
    t = timed(b, 20)
    print(f"batch of {k}: untuned median {u[0]:.1f}  tuned median {t[0]:.1f} us", flush=True)
