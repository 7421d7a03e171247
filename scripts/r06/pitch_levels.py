#!/usr/bin/env python3
"""Batch of NB x 8192^2, J = 5, forward (and inverse): per-level kernel times and the call's time for several row pitches of
the source / destination images -- dense (32768 B), libdwt's dwt_util_get_stride pitch (33344 B), and others."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, J = 8192, 5
nb = int(os.environ.get("NB", 16))
for pad in [int(x) for x in os.environ.get("PADS", "0,144,64,32,256,1024").split(",")]:
    p = n + pad
    src = torch.rand((nb, n, p), device="cuda"); dst = torch.empty_like(src); rec = torch.empty_like(src)
    for inverse, a, b in ((0, src, dst), (1, dst, rec)):
        fn = lambda: dwt.transform2d_batch("cdf97_s", inverse, a, b, n * p * 4, nb, p * 4, n, n, J)
        for _ in range(3): fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
        for s, e in ev:
            s.record(); fn(); e.record()
        torch.cuda.synchronize()
        t = statistics.median(s.elapsed_time(e) for s, e in ev)
        dwt.prof_enable(2)
        for _ in range(6): fn()
        torch.cuda.synchronize()
        ms, cnt = dwt.prof_read_levels(J); dwt.prof_enable(0)
        gs = nb * n * n / t / 1e6
        print(f"pitch {p * 4:6d} B {'inv' if inverse else 'fwd'}: call {t:7.3f} ms = {gs:6.1f} Gsamples/s | levels (us): " + " ".join(f"L{j}:{ms[j] * 1e3:.0f}" for j in range(J)), flush=True)
    del src, dst, rec
