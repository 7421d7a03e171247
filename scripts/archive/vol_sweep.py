#!/usr/bin/env python3
"""A/B the 3-D path's knobs in one process: python scripts/vol_sweep.py [n] [levels]
Each line: option set, median ms of 7 calls (forward, in place)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lv = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dwt.dwt_util_init(); dwt.use_torch_stream()
V = torch.rand((n, n, n), device="cuda")
DEFAULTS = dict(cpt=0, tile_pairs=0, waves=4, ring=0, wave_horiz=-1, xcd_swizzle=1, vol_cpt=8, vol_tile_pairs=0, vol_nt=-1)
VARIANTS = [
    "", "vol_nt=1", "vol_nt=2", "vol_nt=3", "vol_cpt=8", "vol_cpt=8,vol_nt=3",
    "vol_tile_pairs=32", "vol_tile_pairs=128", "vol_tile_pairs=256", "vol_tile_pairs=512",
    "ring=16", "ring=16,wave_horiz=0", "ring=8,wave_horiz=1", "tile_pairs=32", "tile_pairs=128", "tile_pairs=256",
    "waves=2", "waves=1", "xcd_swizzle=0", "",
]
if os.environ.get("VARIANTS"):
    VARIANTS = os.environ["VARIANTS"].split(";")
def run(opts):
    for k, v in DEFAULTS.items():
        dwt.set_option(k, v)
    for kv in [x for x in opts.split(",") if x]:
        k, v = kv.split("=")
        dwt.set_option(k, int(v))
    ts = []
    for i in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dwt.transform3d(0, V, n * 4, n * n * 4, n, n, n, lv)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2], ts[0]
for o in VARIANTS:
    med, mn = run(o)
    print(f"{o or 'default':34s} median {med*1e3:8.3f} ms  min {mn*1e3:8.3f} ms  {n**3/med/1e9:7.1f} Gvox/s", flush=True)
