#!/usr/bin/env python3
"""Does the row pitch (not the row length) set the bandwidth?  One forward level over a batch
of n x n images stored with different row pitches: python scripts/pitch_probe.py [n] [images]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dwt.dwt_util_init(); dwt.use_torch_stream()
for pitch_elems in (n, n + 64, n + 16, 2 * n, 2 * n + 64, 4 * n, 8 * n, 8 * n + 64):
    if nb * n * pitch_elems * 4 * 2 > 60e9:
        continue
    src = torch.rand((nb, n, pitch_elems), device="cuda"); dst = torch.empty_like(src)
    fn = lambda: dwt.transform2d_batch("cdf97_s", 0, src, dst, n * pitch_elems * 4, nb, pitch_elems * 4, n, n, 1)
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(7):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = statistics.median(ts)
    print(f"{nb} x {n}x{n}, pitch {pitch_elems*4:6d} B: {t*1e3:7.3f} ms  {2*4*n*n*nb/t/1e9:7.1f} GB/s useful", flush=True)
    del src, dst
