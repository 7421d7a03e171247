#!/usr/bin/env python3
"""1024^3 (or SIZE^3): the 3-D entries timed with HIP events -- out of place 1 / 3 levels, in place forward / inverse 1 / 3 levels."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n = int(os.environ.get("SIZE", 1024))
a = torch.rand((n, n, n), device="cuda"); b = torch.empty_like(a)
def timed(fn, reps=7):
    for _ in range(2): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = [s.elapsed_time(e) for s, e in ev]
    return statistics.median(t), min(t)
for name, fn in (("op 1 level", lambda: dwt.transform3d_op(a, b, n * 4, n * n * 4, n, n, n, 1)),
                 ("op 3 levels", lambda: dwt.transform3d_op(a, b, n * 4, n * n * 4, n, n, n, 3)),
                 ("ip fwd 1", lambda: dwt.transform3d(0, b, n * 4, n * n * 4, n, n, n, 1)),
                 ("ip inv 1", lambda: dwt.transform3d(1, b, n * 4, n * n * 4, n, n, n, 1)),
                 ("ip fwd 3", lambda: dwt.transform3d(0, b, n * 4, n * n * 4, n, n, n, 3)),
                 ("ip inv 3", lambda: dwt.transform3d(1, b, n * 4, n * n * 4, n, n, n, 3))):
    med, mn = timed(fn)
    print(f"{name:12s} median {med:7.3f} ms  min {mn:7.3f}", flush=True)
