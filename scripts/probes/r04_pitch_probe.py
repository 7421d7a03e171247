#!/usr/bin/env python3
"""Fast / slow placement (DESIGN s5), round 4: is it the POWER-OF-TWO PITCH?  For several physical
placements of the batch's two buffers (re-allocated behind spacers), the level-0 rate and the whole
5-level step for row pitches 32768 B + pad, image strides + extra, and padded LL scratch pitches --
all views into the SAME two allocations, so each row of the table is one physical placement."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, nb = 8192, int(os.environ.get("IMAGES", 64))
PADS = [0, 16, 32, 64, 128, 256, 512, 1024]           # elements (x4 bytes)
EXTRA = [0, 64 << 10, (2 << 20) + (64 << 10)]        # bytes added to the image stride (pad 0)
PADMAX = max(PADS)
dwt.dwt_util_init(); dwt.use_torch_stream()
raw_elems = nb * (n * (n + PADMAX)) + nb * max(EXTRA) // 4 + 1024
def rate(src, dst, pitch, bstride, J, reps=5):
    for _ in range(2): dwt.transform2d_batch("cdf97_s", 0, src, dst, bstride, nb, pitch, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(True)
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): dwt.transform2d_batch("cdf97_s", 0, src, dst, bstride, nb, pitch, n, n, J)
    b.record(); torch.cuda.synchronize(); ms, cnt = dwt.prof_read(); dwt.prof_enable(False)
    return 2 * 4 * n * n * nb / (ms / cnt) / 1e6, a.elapsed_time(b) / reps
hold = []
for k, sp in enumerate([0, 1.3, 7, 2.6, 11, 23][:int(os.environ.get("PLACEMENTS", 6))]):
    spacer = torch.empty(int(sp * (1 << 30)), dtype=torch.uint8, device="cuda") if sp else None
    src = torch.rand(raw_elems, device="cuda"); dst = torch.empty_like(src)
    print(f"== placement {k} (spacer {sp} GiB) src {src.data_ptr():#x} dst {dst.data_ptr():#x}", flush=True)
    for pad in PADS:
        pitch = (n + pad) * 4
        r1, _ = rate(src, dst, pitch, pitch * n, 1)
        dwt.set_option("ll_pad", 0)
        r5, t5 = rate(src, dst, pitch, pitch * n, 5)
        dwt.set_option("ll_pad", 64)
        r5p, t5p = rate(src, dst, pitch, pitch * n, 5)
        dwt.set_option("ll_pad", 0)
        print(f"  pitch 32768+{pad*4:5d} B: J=1 level0 {r1:6.0f} GB/s | J=5 level0 {r5:6.0f} GB/s step {t5:7.3f} ms = {nb*n*n/t5/1e6:6.1f} Gs/s | ll_pad 64: level0 {r5p:6.0f} step {t5p:7.3f} ms", flush=True)
    for ex in EXTRA[1:]:
        r1, _ = rate(src, dst, n * 4, n * n * 4 + ex, 1)
        r5, t5 = rate(src, dst, n * 4, n * n * 4 + ex, 5)
        print(f"  pitch 32768, image stride +{ex:8d} B: J=1 level0 {r1:6.0f} | J=5 level0 {r5:6.0f} step {t5:7.3f} ms", flush=True)
    hold.append(spacer)
    del src, dst; torch.cuda.empty_cache()
