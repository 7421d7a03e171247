#!/usr/bin/env python3
"""Double-precision 9/7 and 5/3, 4096^2, J = 3, device resident: forward / inverse calls on one image and a batch of 8
(HIP-event medians), with the library named by DWT_HIP_LIB."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, J, nb = 4096, 3, 8
img = torch.rand((nb, n, n), device="cuda", dtype=torch.float64); out = torch.empty_like(img)
def timed(fn, reps=30):
    for i in range(4): fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(i); b.record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) * 1e3 for a, b in ev]
    return statistics.median(t), min(t)
print(os.environ.get("DWT_HIP_LIB", "(shipped build)"))
for w in ("cdf97_d", "cdf53_d"):
    for name, fn in ((w + " fwd x8", lambda i: dwt.transform2d_batch(w, 0, img, out, n * n * 8, nb, n * 8, n, n, J)),
                     (w + " inv x8", lambda i: dwt.transform2d_batch(w, 1, img, out, n * n * 8, nb, n * 8, n, n, J)),
                     (w + " fwd x1", lambda i: dwt.transform2d_batch(w, 0, img[i % nb], out[i % nb], n * n * 8, 1, n * 8, n, n, J))):
        med, mn = timed(fn)
        print(f"    {name:16s} median {med:8.1f} us  min {mn:8.1f}", flush=True)
