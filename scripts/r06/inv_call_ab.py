#!/usr/bin/env python3
"""Whole inverse calls (J = 5) under option settings, alternated in one process: one 8192^2 image through
dwt_cdf97_2i_s2 and dwt_cdf97_2i_s (rotating over 8 images), batches of 8 and 32.  Usage: inv_call_ab.py "k=v,k=v" ..."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, J = int(os.environ.get("SIZE", 8192)), 5
variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in sys.argv[1:]] or [{}]
keys = sorted({k for v in variants for k in v})
defaults = {k: dwt.get_option(k) for k in keys}
nb = 8
img = torch.rand((nb, n, n), device="cuda"); co = torch.empty_like(img); out = torch.empty_like(img)
dwt.transform2d_batch("cdf97_s", 0, img, co, n * n * 4, nb, n * 4, n, n, J)
work = co.clone()
def timed(fn, reps):
    for i in range(4): fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(i); b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 for a, b in ev]
legs = {
    "2i_s2 x1": (lambda i: dwt.dwt_cdf97_2i_s2(co[i % nb], out[i % nb], n * 4, 4, n, n, n, n, J), 40),
    "2i_s  x1": (lambda i: dwt.dwt_cdf97_2i_s(work[i % nb], n * 4, 4, n, n, n, n, J), 40),
    "batch x8": (lambda i: dwt.transform2d_batch("cdf97_s", 1, co, out, n * n * 4, nb, n * 4, n, n, J), 10),
}
res = {(l, vi): [] for l in legs for vi in range(len(variants))}
for rnd in range(4):
    for vi, v in enumerate(variants):
        for k in keys:
            dwt.set_option(k, int(v.get(k, defaults[k])))
        for l, (fn, reps) in legs.items():
            if l.startswith("2i_s "):
                work.copy_(co)
            t = timed(fn, reps)
            if rnd:
                res[(l, vi)] += t
for k in keys:
    dwt.set_option(k, defaults[k])
for l in legs:
    for vi, v in enumerate(variants):
        t = res[(l, vi)]
        print(f"{l}  {sys.argv[1 + vi] if len(sys.argv) > 1 else 'default':40s} median {statistics.median(t):8.1f} us  min {min(t):8.1f}", flush=True)
