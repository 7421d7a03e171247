#!/usr/bin/env python3
"""Round 4: PLACED allocation (dwt_hip_alloc_placed: source + LL scratch from one physical class, the
destination from another) against plain hipMalloc, 64-image batch: level-0 / step rates."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, nb = 8192, int(os.environ.get("IMAGES", 64))
dwt.dwt_util_init(); dwt.use_torch_stream()
L = dwt.lib
def rate(src, dst, J, pitch, reps=4):
    bs = pitch * n
    for _ in range(2): dwt.transform2d_batch("cdf97_s", 0, src, dst, bs, nb, pitch, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(2)
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(reps): dwt.transform2d_batch("cdf97_s", 0, src, dst, bs, nb, pitch, n, n, J)
    b.record(); torch.cuda.synchronize(); ms, cnt = dwt.prof_read_levels(6); dwt.prof_enable(False)
    return 2 * 4 * n * n * nb / ms[0] / 1e6, a.elapsed_time(b) / reps, [round(m * 1e3) for m in ms[:5]]
b0, b1 = nb * (n // 2) ** 2 * 4 + 4096, nb * (n // 4) ** 2 * 4 + 4096
big = nb * n * (n + 256) * 4
for mode in os.environ.get("MODES", "placed1,placed2,plain,placed1,placed2,plain").split(","):
    t0 = time.time()
    if mode.startswith("placed"):
        dwt.set_option("placed_prefer", int(mode[6:] or 0))
        (src, w0, w1), (dst,), stats = dwt.alloc_placed([big, b0, b1], [big])
    else:
        src, dst, w0, w1 = [L.dwt_hip_malloc(x) for x in (big, big, b0, b1)]; stats = {}
    t_alloc = time.time() - t0
    L.dwt_hip_probe_pair_us(src, dst, big)  # finite data in both
    assert L.dwt_hip_set_workspace(w0, b0, w1, b1) == 0, dwt.last_error()
    out = []
    for pitch in (n * 4, n * 4 + 1024):
        r1, _, _ = rate(src, dst, 1, pitch)
        r5, t5, lv = rate(src, dst, 5, pitch)
        out.append(f"pitch {pitch}: level0 J=1 {r1:5.0f} J=5 {r5:5.0f} GB/s step {t5:6.3f} ms = {nb*n*n/t5/1e6:6.1f} Gs/s levels {lv} us")
    print(f"{mode:6s}: alloc {t_alloc:5.1f} s {stats} | " + " | ".join(out), flush=True)
    L.dwt_hip_set_workspace(None, 0, None, 0)
    for p in (src, dst, w0, w1):
        if mode.startswith("placed"): L.dwt_hip_free_mapped(p)
        else: L.dwt_hip_free(p)
