#!/usr/bin/env python3
"""Batch of 8 images of 8192^2 (the shard of one GPU in the 8-GPU split): option variants A/B in one
process.  python scripts/batch8_variants.py "" "pipeline=2" ..."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
nb, n, J = int(os.environ.get("IMAGES", 8)), 8192, 5
inv = int(os.environ.get("INVERSE", 0))
dwt.dwt_util_init(); dwt.use_torch_stream()
x = torch.rand((nb, n, n), device="cuda"); y = torch.empty_like(x)
DEF = dict(pipeline=0, tile_pairs=0, ring=0, cpt=0, ring_inv=0, waves=4, wave_horiz_inv=-1, nt=7)
for v in sys.argv[1:] or [""]:
    opts = dict(DEF)
    for kv in [t for t in v.split(",") if t]:
        k, val = kv.split("="); opts[k] = int(val)
    for k, val in opts.items(): dwt.set_option(k, val)
    fn = lambda: dwt.transform2d_batch("cdf97_s", inv, x, y, n * n * 4, nb, n * 4, n, n, J)
    for _ in range(4): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(15):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    t = statistics.median(ts)
    print(f"{v:34s} {'inverse' if inv else 'forward'} {nb} images: {t:6.3f} ms  {nb*n*n/t/1e6:6.1f} Gsamples/s", flush=True)
