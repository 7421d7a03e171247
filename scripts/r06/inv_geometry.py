#!/usr/bin/env python3
"""Inverse level 0 (one level, dwt_cdf97_2i_s2 semantics) under (columns per lane, tile height) variants for a list of
shapes: one image (rotating over enough images to stay out of the 256 MiB Infinity Cache) and batches.  Alternated in
one process, medians of HIP-event times.  Usage: inv_geometry.py [shape ...] with shape = WxHxB; env VARIANTS."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
J = int(os.environ.get("LEVELS", 1))
INV = int(os.environ.get("INVERSE", 1))
shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]] or [(8192, 8192, 1), (8192, 8192, 8), (8192, 4096, 1), (16384, 8192, 1), (7000, 5000, 1), (4096, 4096, 1)]
variants = [tuple(int(x) for x in v.split(":")) for v in os.environ.get("VARIANTS", "0:0,4:32,4:16,4:64,8:64,8:32,8:16").split(",")]
for (w, h, b) in shapes:
    per = w * h * 4 * b
    nrot = max(2, min(16, (1 << 30) // per + 1)) if b * w * h * 4 < (1 << 30) else 2
    src = torch.rand((nrot, b, h, w), device="cuda"); dst = torch.empty_like(src)
    if INV:
        for r in range(nrot):
            dwt.transform2d_batch("cdf97_s", 0, src[r], dst[r], w * h * 4, b, w * 4, w, h, J)
        src, dst = dst, src
    res = {v: [] for v in variants}
    for rnd in range(4):
        for v in variants:
            dwt.set_option("cpt", v[0]); dwt.set_option("tile_pairs", v[1])
            fn = lambda i: dwt.transform2d_batch("cdf97_s", INV, src[i % nrot], dst[i % nrot], w * h * 4, b, w * 4, w, h, J)
            for i in range(3): fn(i)
            reps = 12 if per < (1 << 30) else 5
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for i, (a, e) in enumerate(ev):
                a.record(); fn(i); e.record()
            torch.cuda.synchronize()
            if rnd:
                res[v] += [a.elapsed_time(e) * 1e3 for a, e in ev]
    dwt.set_option("cpt", 0); dwt.set_option("tile_pairs", 0)
    alg = 0
    for j in range(J):
        alg += 2 * 4 * (-(-w // (1 << j))) * (-(-h // (1 << j))) * b
    print(f"{'inverse' if INV else 'forward'} {w}x{h} x{b} J={J} ({nrot} rotating buffers):", flush=True)
    for v in variants:
        med = statistics.median(res[v])
        print(f"    cpt {v[0]} tile_pairs {v[1]:3d}: median {med:8.1f} us  min {min(res[v]):8.1f}   {alg / med / 1e3:7.1f} GB/s", flush=True)
    del src, dst
    torch.cuda.empty_cache()
