#!/usr/bin/env python3
"""Inverse level 0 of a batch (default 32 x 8192^2) under every knob the inverse sweep has: tile height, columns per
lane, ring depth, waves per workgroup, XCD swizzle.  Alternated, medians of HIP-event times, two rotating batches."""
import itertools, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, b = int(os.environ.get("SIZE", 8192)), int(os.environ.get("IMAGES", 32))
src = torch.rand((2, b, n, n), device="cuda"); dst = torch.empty_like(src)
for r in range(2):
    dwt.transform2d_batch("cdf97_s", 0, src[r], dst[r], n * n * 4, b, n * 4, n, n, 1)
src, dst = dst, src
keys = ("tile_pairs", "cpt", "ring_inv", "waves", "xcd_swizzle")
base = {k: dwt.get_option(k) for k in keys}
variants = [dict(zip(keys, v)) for v in itertools.product((16, 32), (4, 8), (8, 16), (4, 2), (1, 0))]
res = {i: [] for i in range(len(variants))}
for rnd in range(3):
    for i, v in enumerate(variants):
        for k in keys:
            dwt.set_option(k, v[k])
        fn = lambda j: dwt.transform2d_batch("cdf97_s", 1, src[j % 2], dst[j % 2], n * n * 4, b, n * 4, n, n, 1)
        fn(0)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
        for j, (a, e) in enumerate(ev):
            a.record(); fn(j); e.record()
        torch.cuda.synchronize()
        if rnd:
            res[i] += [a.elapsed_time(e) * 1e3 for a, e in ev]
for k in keys:
    dwt.set_option(k, base[k])
rows = sorted((statistics.median(res[i]), i) for i in res)
for med, i in rows:
    print(f"{variants[i]}: {med:8.1f} us  {2 * 4 * n * n * b / med / 1e3:7.1f} GB/s", flush=True)
