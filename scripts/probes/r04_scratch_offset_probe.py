#!/usr/bin/env python3
"""Fast / slow placement, round 4: the level-0 rate against the OFFSET of the LL scratch band inside
its allocation, on a slow and on a fast placement of the batch (one process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, nb, J = 8192, int(os.environ.get("IMAGES", 64)), int(os.environ.get("LEVELS", 2))
img = n * n * 4
dwt.dwt_util_init(); dwt.use_torch_stream()
OFFS_KIB = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576,
            3, 5, 12, 20, 48, 80, 192, 320, 768, 1280, 3072, 5120, 1024 + 64, 2048 + 4, 4096 + 256]
dwt.set_option("ll_offset_kib", max(OFFS_KIB))  # one allocation big enough for every offset
def l0(src, dst, reps=4):
    for _ in range(2): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n*4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(2)
    for _ in range(reps): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n*4, n, n, J)
    torch.cuda.synchronize(); ms, cnt = dwt.prof_read_levels(4); dwt.prof_enable(False)
    return 2*4*n*n*nb/ms[0]/1e6, 2*4*(n//2)**2*nb/ms[1]/1e6
src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)
l0(src, dst)
found, hold = {}, []
for k, sp in enumerate([0, 1.3, 7, 2.6, 11, 23, 3.3, 5.1]):
    spacer = torch.empty(int(sp * (1 << 30)), dtype=torch.uint8, device="cuda") if sp else None
    if k: src = torch.rand((nb, n, n), device="cuda"); dst = torch.empty_like(src)
    dwt.set_option("ll_offset_kib", 0)
    r, r1 = l0(src, dst)
    kind = "slow" if r < 5600 else "fast" if r > 5950 else None
    print(f"placement {k} (spacer {sp} GiB): level0 {r:.0f} level1 {r1:.0f} GB/s {kind or ''}", flush=True)
    if kind and kind not in found:
        found[kind] = (src, dst); hold.append(spacer)
    else:
        del src, dst, spacer; torch.cuda.empty_cache()
    if len(found) == 2: break
for off in OFFS_KIB:
    dwt.set_option("ll_offset_kib", off)
    print(f"ll offset {off:8d} KiB: " + "   ".join(f"{kind}: level0 %6.0f level1 %6.0f" % l0(*found[kind]) for kind in ("fast", "slow") if kind in found), flush=True)
