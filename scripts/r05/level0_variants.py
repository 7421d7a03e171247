#!/usr/bin/env python3
"""Level 0 of ONE 8192^2 image under (cpt, tile_pairs, ring, waves) variants, forward and inverse; then the per-level
times of the whole five-level call."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, nb = 8192, 8
src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)

def timed(fn, reps=30):
    for i in range(5): fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(i); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[0]

for inverse in (0, 1):
    rows = []
    for cpt, tp, ring, waves in itertools.product((0,) if inverse else (8, 4), (0, 128, 64, 32, 16), (8, 16), (4, 2)):
        dwt.set_option("cpt", cpt); dwt.set_option("tile_pairs", tp); dwt.set_option("waves", waves)
        dwt.set_option("ring_inv" if inverse else "ring", ring)
        if inverse:
            fn = lambda i: dwt.dwt_cdf97_2i_s2(src[i % nb], dst[i % nb], n * 4, 4, n, n, n, n, 1)
        else:
            fn = lambda i: dwt.dwt_cdf97_2f_s2(src[i % nb], dst[i % nb], n * 4, 4, n, n, n, n, 1)
        med, mn = timed(fn)
        rows.append((med, mn, cpt, tp, ring, waves))
    for k, v in (("cpt", 0), ("tile_pairs", 0), ("ring", 0), ("ring_inv", 8), ("waves", 4)):
        dwt.set_option(k, v)
    rows.sort()
    print("inverse" if inverse else "forward", "8192^2 one level:", flush=True)
    for r in rows[:8] + [r for r in rows if r[3] == 0 and r[5] == 4]:
        print(f"    cpt {r[2]} tile_pairs {r[3]:3d} ring {r[4]:2d} waves {r[5]}: median {r[0]:6.1f} us  min {r[1]:6.1f}", flush=True)
J = 5
for inverse in (0, 1):
    if inverse:
        dwt.transform2d_batch("cdf97_s", 0, src, dst, n * n * 4, nb, n * 4, n, n, J)
        fn = lambda i: dwt.dwt_cdf97_2i_s2(dst[i % nb], src[i % nb], n * 4, 4, n, n, n, n, J)
    else:
        fn = lambda i: dwt.dwt_cdf97_2f_s2(src[i % nb], dst[i % nb], n * 4, 4, n, n, n, n, J)
    med, mn = timed(fn, 50)
    dwt.prof_enable(2)
    for i in range(20): fn(i)
    torch.cuda.synchronize()
    ms, cnt = dwt.prof_read_levels(J); dwt.prof_enable(0)
    print(("inverse" if inverse else "forward"), f"J=5 call: median {med:.1f} min {mn:.1f} us; levels (with event overhead): " + " ".join(f"L{j}:{ms[j]*1e3:.1f}" for j in range(J)), flush=True)
