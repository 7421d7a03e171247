#!/usr/bin/env python3
"""Interleaved-layout entries at 8192^2 J=5 (device resident): python scripts/il_bench.py [reps]
Forward with the exact border strips on the side stream (il_lazy_strips=1, default) and in line (0)."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dwt.dwt_util_init(); dwt.use_torch_stream()
n, J = 8192, 5
a = torch.rand((n, n), device="cuda"); b = torch.empty_like(a); c = a.clone()
def t(name, fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{name:40s} median {statistics.median(ts)*1e6:8.1f} us  min {min(ts)*1e6:8.1f} us", flush=True)
for rnd in range(2):
    for lazy in (1, 0):
        dwt.set_option("il_lazy_strips", lazy)
        t(f"fwd out-of-place  lazy_strips={lazy}", lambda: dwt.transform2d_interleaved("cdf97_s", 0, 0, a, b, n*4, 4, n, n, None, None, J))
        t(f"fwd in-place      lazy_strips={lazy}", lambda: dwt.dwt_cdf97_2f_inplace_s(c, n*4, 4, n, n, n, n, J))
dwt.set_option("il_lazy_strips", 1)
# same bits either way
x = torch.empty_like(a); y = torch.empty_like(a)
dwt.transform2d_interleaved("cdf97_s", 0, 0, a, x, n*4, 4, n, n, None, None, J)
dwt.set_option("il_lazy_strips", 0)
dwt.transform2d_interleaved("cdf97_s", 0, 0, a, y, n*4, 4, n, n, None, None, J)
dwt.set_option("il_lazy_strips", 1)
torch.cuda.synchronize()
print("lazy == in line:", torch.equal(x, y), flush=True)
t("inv out-of-place", lambda: dwt.transform2d_interleaved("cdf97_s", 1, 0, b, c, n*4, 4, n, n, None, None, J))
t("inv in-place", lambda: dwt.dwt_cdf97_2i_inplace_s(c, n*4, 4, n, n, n, n, J))
