#!/usr/bin/env python3
"""Single-image latency of the INVERSE entries, per level, under option variants:
    python scripts/single_levels_inv.py "" "tile_pairs=16" ...
(see scripts/single_levels.py for the forward counterpart)"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt

DEFAULTS = dict(generic=0, cpt=0, tile_pairs=0, waves=4, xcd_swizzle=1, ring_inv=8)
n = int(os.environ.get("SIZE", 8192)); J = int(os.environ.get("LEVELS", 5)); nb = 8; reps = int(os.environ.get("REPS", 40))
dwt.dwt_util_init(); dwt.use_torch_stream()
src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)

def timed(fn):
    for i in range(8): fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(i); b.record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) * 1e3 for a, b in ev]
    return statistics.median(t), min(t)

for v in sys.argv[1:] or [""]:
    opts = dict(DEFAULTS)
    for kv in [x for x in v.split(",") if x]:
        k, val = kv.split("="); opts[k] = int(val)
    for k, val in opts.items(): dwt.set_option(k, val)
    s2 = lambda i: dwt.dwt_cdf97_2i_s2(src[i % nb], dst[i % nb], n * 4, 4, n, n, n, n, J)
    ip = lambda i: dwt.dwt_cdf97_2i_s(dst[i % nb], n * 4, 4, n, n, n, n, J)
    med2, min2 = timed(s2)
    medi, mini = timed(ip)
    dwt.prof_enable(2)
    for i in range(20): s2(i)
    torch.cuda.synchronize()
    ms, cnt = dwt.prof_read_levels(J); dwt.prof_enable(0)
    lv = " ".join(f"L{j}:{ms[j]*1e3:6.1f}" for j in range(J))
    print(f"{v:34s} inverse s2 {med2:6.1f} (min {min2:6.1f})  inplace {medi:6.1f} (min {mini:6.1f}) us | levels(us, with event overhead) {lv}", flush=True)
