#!/usr/bin/env python3
"""One level of a single image under every (cpt, tile_pairs, ring) variant: what could a wider tuner find?
    python scripts/r05/level_variants.py [fwd|inv]
Each shape: a one-level out-of-place call on 8 rotating images (nothing cached), HIP events, median of 30."""
import os, sys, statistics, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt

inverse = len(sys.argv) > 1 and sys.argv[1] == "inv"
dwt.dwt_util_init(); dwt.use_torch_stream()

def timed(fn, reps=30):
    for i in range(5): fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(i); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[0]

for n in (4096, 2048, 1024, 512):
    nb = 8 if n >= 2048 else 32
    src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)
    rows = []
    cpts = (0,) if inverse else ((8, 4) if n >= 1024 else (4,))
    for cpt, tp, ring in itertools.product(cpts, (0, 64, 32, 16, 8, 4, 2, 1), (8, 16)):
        if tp > n // 2 // 4 and tp: continue
        dwt.set_option("cpt", cpt); dwt.set_option("tile_pairs", tp)
        dwt.set_option("ring_inv" if inverse else "ring", ring)
        if inverse:
            fn = lambda i: dwt.dwt_cdf97_2i_s2(src[i % nb], dst[i % nb], n * 4, 4, n, n, n, n, 1)
        else:
            fn = lambda i: dwt.dwt_cdf97_2f_s2(src[i % nb], dst[i % nb], n * 4, 4, n, n, n, n, 1)
        try:
            med, mn = timed(fn)
        except Exception as e:
            print("   failed", cpt, tp, ring, e); continue
        rows.append((med, mn, cpt, tp, ring))
    dwt.set_option("cpt", 0); dwt.set_option("tile_pairs", 0); dwt.set_option("ring", 0); dwt.set_option("ring_inv", 8)
    base = [r for r in rows if r[3] == 0]
    rows.sort()
    print(f"{'inv' if inverse else 'fwd'} {n}^2: defaults " + ", ".join(f"cpt{r[2]} ring{r[4]}: {r[0]:.1f}" for r in base), flush=True)
    for r in rows[:6]:
        print(f"    cpt {r[2]} tile_pairs {r[3]:3d} ring {r[4]:2d}: median {r[0]:6.1f} us  min {r[1]:6.1f}", flush=True)
    del src, dst
