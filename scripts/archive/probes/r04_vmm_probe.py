#!/usr/bin/env python3
"""Round 4: buffers mapped from physical pieces that come from far-apart physical regions, interleaved
at piece granularity (dwt_hip_malloc_mapped) -- single-stream write rate of such a buffer and the level-0
rate of the 64-image batch with source, destination and LL scratch all allocated that way."""
import os, sys, time
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, nb = 8192, int(os.environ.get("IMAGES", 64))
img = n * n * 4
GB, MB = 1 << 30, 1 << 20
dwt.dwt_util_init(); dwt.use_torch_stream()
L = dwt.lib
def rate(src, dst, J, reps=4):
    for _ in range(2): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n * 4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(2)
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(reps): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n * 4, n, n, J)
    b.record(); torch.cuda.synchronize(); ms, cnt = dwt.prof_read_levels(6); dwt.prof_enable(False)
    return 2 * 4 * n * n * nb / ms[0] / 1e6, a.elapsed_time(b) / reps
b0, b1 = nb * (n // 2) ** 2 * 4 + 4096, nb * (n // 4) ** 2 * 4 + 4096
CONFIGS = [(2, 1, 0), (2, 8, 15), (2, 16, 7), (2, 24, 5), (32, 8, 15), (2, 3, 30), (2, 6, 16)]
for piece_mb, slices, ballast_gb in CONFIGS:
    t0 = time.time()
    bufs = []
    for nbytes in (nb * img, nb * img, b0, b1):
        p = L.dwt_hip_malloc_mapped(nbytes, piece_mb * MB, slices, int(max(0, ballast_gb * GB - nbytes // max(slices, 1))) if slices > 1 else 0)
        if not p:
            print("alloc failed:", dwt.last_error()); break
        bufs.append(p)
    if len(bufs) < 4:
        for p in bufs: L.dwt_hip_free_mapped(p)
        continue
    t_alloc = time.time() - t0
    src, dst, w0, w1 = bufs
    one = nb * img / L.dwt_hip_probe_pair_us(src, None, nb * img) / 1e3
    L.dwt_hip_probe_pair_us(dst, None, nb * img)
    assert L.dwt_hip_set_workspace(w0, b0, w1, b1) == 0, dwt.last_error()
    try:
        r1, _ = rate(src, dst, 1)
        r2, _ = rate(src, dst, 2)
        r5, t5 = rate(src, dst, 5)
        print(f"pieces {piece_mb:3d} MiB x {slices:2d} slices, ballast {ballast_gb:2d} GiB: alloc {t_alloc:5.1f} s | single-stream write {one:5.0f} GB/s | "
              f"level0 J=1 {r1:5.0f}  J=2 {r2:5.0f}  J=5 {r5:5.0f} GB/s, step {t5:6.3f} ms = {nb*n*n/t5/1e6:6.1f} Gs/s", flush=True)
    except Exception as e:
        print("transform failed:", e, flush=True)
    L.dwt_hip_set_workspace(None, 0, None, 0)
    t0 = time.time()
    for p in bufs: L.dwt_hip_free_mapped(p)
    print(f"   free {time.time() - t0:.1f} s", flush=True)
