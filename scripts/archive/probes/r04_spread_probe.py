#!/usr/bin/env python3
"""Round 4: SPREAD allocation (dwt_hip_malloc_spread: pieces at even distances through all free
physical memory) -- single-stream write rate and the 64-image batch's level-0 / step rates with source,
destination and LL scratch allocated that way, per piece size; plain hipMalloc beside it."""
import os, sys, time
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, nb = 8192, int(os.environ.get("IMAGES", 64))
GB, MB = 1 << 30, 1 << 20
dwt.dwt_util_init(); dwt.use_torch_stream()
L = dwt.lib
def rate(src, dst, J, pitch, reps=4):
    bs = pitch * n
    for _ in range(2): dwt.transform2d_batch("cdf97_s", 0, src, dst, bs, nb, pitch, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(2)
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(reps): dwt.transform2d_batch("cdf97_s", 0, src, dst, bs, nb, pitch, n, n, J)
    b.record(); torch.cuda.synchronize(); ms, cnt = dwt.prof_read_levels(6); dwt.prof_enable(False)
    return 2 * 4 * n * n * nb / ms[0] / 1e6, a.elapsed_time(b) / reps
b0, b1 = nb * (n // 2) ** 2 * 4 + 4096, nb * (n // 4) ** 2 * 4 + 4096
big = nb * n * (n + 256) * 4
for piece_mb in [int(x) for x in os.environ.get("PIECES", "2,8,32,2,0,0").split(",")]:
    t0 = time.time()
    bufs = []
    for nbytes in (big, big, b0, b1):
        p = L.dwt_hip_malloc_spread(nbytes, piece_mb * MB) if piece_mb else L.dwt_hip_malloc(nbytes)
        if not p:
            print("alloc failed:", dwt.last_error()); break
        bufs.append(p)
    if len(bufs) < 4:
        break
    t_alloc = time.time() - t0
    src, dst, w0, w1 = bufs
    one = big / L.dwt_hip_probe_pair_us(src, None, big) / 1e3
    L.dwt_hip_probe_pair_us(dst, None, big)
    assert L.dwt_hip_set_workspace(w0, b0, w1, b1) == 0, dwt.last_error()
    out = []
    for pitch in (n * 4, n * 4 + 1024):
        r1, _ = rate(src, dst, 1, pitch)
        r5, t5 = rate(src, dst, 5, pitch)
        out.append(f"pitch {pitch}: level0 J=1 {r1:5.0f} J=5 {r5:5.0f} GB/s step {t5:6.3f} ms = {nb*n*n/t5/1e6:6.1f} Gs/s")
    print(f"{'spread, pieces %2d MiB' % piece_mb if piece_mb else 'plain hipMalloc       '}: alloc {t_alloc:5.1f} s | single-stream write {one:5.0f} GB/s | " + " | ".join(out), flush=True)
    L.dwt_hip_set_workspace(None, 0, None, 0)
    for p in bufs:
        if piece_mb: L.dwt_hip_free_mapped(p)
        else: L.dwt_hip_free(p)
