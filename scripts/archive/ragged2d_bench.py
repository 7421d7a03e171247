#!/usr/bin/env python3
"""2-D forward / inverse float 9/7 on image sizes that are not made of whole tiles, beside 8192^2.
python scripts/ragged2d_bench.py"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
J = 5
for (w, h, nb) in ((8192, 8192, 8), (8000, 8000, 8), (8188, 8192, 8), (7680, 4320, 16), (3840, 2160, 64), (1920, 1080, 256), (4000, 3000, 32), (4001, 3001, 32)):
    x = torch.rand((nb, h, w), device="cuda"); y = torch.empty_like(x)
    for inv in (0, 1):
        fn = lambda: dwt.transform2d_batch("cdf97_s", inv, x, y, w * h * 4, nb, w * 4, w, h, J)
        for _ in range(3): fn()
        torch.cuda.synchronize(); ts = []
        for _ in range(10):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        t = statistics.median(ts)
        print(f"{w}x{h} x{nb} {'inverse' if inv else 'forward'}: {t:7.3f} ms  {nb*w*h/t/1e6:6.1f} Gsamples/s", flush=True)
    del x, y
