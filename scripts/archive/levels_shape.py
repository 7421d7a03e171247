#!/usr/bin/env python3
"""Per-level kernel time and achieved bandwidth of the batched forward transform for a given shape:
    python scripts/levels_shape.py W H BATCH [LEVELS] ["opt=val,..."]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
w, h, nb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
J = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dwt.dwt_util_init(); dwt.use_torch_stream()
for kv in [x for x in (sys.argv[5] if len(sys.argv) > 5 else "").split(",") if x]:
    k, v = kv.split("="); dwt.set_option(k, int(v))
x = torch.rand((nb, h, w), device="cuda"); y = torch.empty_like(x)
inv = int(os.environ.get("INVERSE", 0))
fn = lambda: dwt.transform2d_batch("cdf97_s", inv, x, y, w * h * 4, nb, w * 4, w, h, J)
for _ in range(3): fn()
dwt.prof_enable(2)
for _ in range(10): fn()
torch.cuda.synchronize()
ms, cnt = dwt.prof_read_levels(J); dwt.prof_enable(0)
tot = 0
for j in range(J):
    wj, hj = -(-w >> j), -(-h >> j)
    b = 8.0 * wj * hj * nb
    tot += ms[j]
    print(f"{w}x{h}x{nb} level {j}: {wj:5d} x {hj:5d}  {ms[j]*1e3:8.1f} us  {b/ms[j]/1e9:6.2f} TB/s")
print(f"sum of levels {tot*1e3:.1f} us -> {nb*w*h/tot/1e6:.1f} Gsamples/s")
