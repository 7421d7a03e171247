#!/usr/bin/env python3
"""Does the rate depend on the stream (hardware queue) the launches go to?  Same 16-image forward
transform timed on the default stream and on several new streams of one process."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
nb, n, J = 16, 8192, 5
dwt.dwt_util_init()
x = torch.rand((nb, n, n), device="cuda"); y = torch.empty_like(x)
torch.cuda.synchronize()
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(9)]
for rnd in range(2):
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            dwt.use_torch_stream()
            for _ in range(3):
                dwt.transform2d_batch("cdf97_s", 0, x, y, n * n * 4, nb, n * 4, n, n, J)
            s.synchronize(); ts = []
            for _ in range(8):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(s); dwt.transform2d_batch("cdf97_s", 0, x, y, n * n * 4, nb, n * 4, n, n, J); e1.record(s)
                e1.synchronize(); ts.append(e0.elapsed_time(e1))
            t = statistics.median(ts)
            print(f"round {rnd} stream {i} ({s.cuda_stream:#x}): {t:6.3f} ms  {nb*n*n/t/1e6:6.1f} Gsamples/s", flush=True)
