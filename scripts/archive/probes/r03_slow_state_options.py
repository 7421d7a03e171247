#!/usr/bin/env python3
"""Fast / slow state (DESIGN s5): find a SLOW and a FAST placement of the batch inside one process, then
run the launch options over both: does any option close the gap?"""
import os, sys, time
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, J, nb = 8192, 1, 64
img = n * n * 4
dwt.dwt_util_init(); dwt.use_torch_stream()
def l0(src, dst, reps=6):
    for _ in range(2): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n*4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(True)
    for _ in range(reps): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n*4, n, n, J)
    torch.cuda.synchronize(); ms, cnt = dwt.prof_read(); dwt.prof_enable(False)
    return 2*4*n*n*nb/(ms/cnt)/1e6
found = {}
hold = []
for k, sp in enumerate([0, 1.3, 0.7, 2.6, 7, 11, 3.3, 5.1, 17, 23]):
    spacer = torch.empty(int(sp * (1 << 30)), dtype=torch.uint8, device="cuda") if sp else None
    src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)
    r = l0(src, dst)
    kind = "slow" if r < 5500 else "fast" if r > 5950 else None
    print(f"placement {k} (spacer {sp} GiB): {r:.0f} GB/s {kind or ''}", flush=True)
    if kind and kind not in found:
        found[kind] = (src, dst); hold.append(spacer)
    else:
        del src, dst, spacer; torch.cuda.empty_cache()
    if len(found) == 2:
        break
D = dict(cpt=0, tile_pairs=0, waves=4, xcd_swizzle=1, ring=0, nt=7, nt_auto=1)
variants = ["", "xcd_swizzle=0", "tile_pairs=32", "tile_pairs=128", "ring=8", "nt=7,nt_auto=0", "nt=1,nt_auto=0", "nt=0,nt_auto=0", "nt=5,nt_auto=0", "cpt=4", "waves=2", "waves=1"]
for v in variants:
    for k_, d_ in D.items(): dwt.set_option(k_, d_)
    for kv in [x for x in v.split(",") if x]:
        k_, val = kv.split("="); dwt.set_option(k_, int(val))
    print(f"{v or 'default':22s} " + "  ".join(f"{kind}: {l0(*found[kind]):6.0f}" for kind in ("fast", "slow") if kind in found) + " GB/s", flush=True)
