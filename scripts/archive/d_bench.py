#!/usr/bin/env python3
"""Double-precision entries (dwt_cdf97_2f_d / _2i_d, dwt_cdf53_2f_d), device resident, in place:
    python scripts/d_bench.py [n] [levels]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
J = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dwt.dwt_util_init(); dwt.use_torch_stream()
a = torch.rand((n, n), device="cuda", dtype=torch.float64)
alg = sum(2 * 8 * (n >> j) ** 2 for j in range(J))
def t(name, fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    m = statistics.median(ts)
    print(f"{name:44s} median {m*1e6:9.1f} us  {n*n/m/1e9:7.1f} Gsamples/s  {alg/m/1e9:7.0f} GB/s algorithmic", flush=True)
t(f"cdf97 double fwd {n}^2 J={J} in place", lambda: dwt.dwt_cdf97_2f_d(a, n * 8, 8, n, n, n, n, J))
t(f"cdf97 double inv {n}^2 J={J} in place", lambda: dwt.dwt_cdf97_2i_d(a, n * 8, 8, n, n, n, n, J))
t(f"cdf53 double fwd {n}^2 J={J} in place", lambda: dwt.dwt_cdf53_2f_d(a, n * 8, 8, n, n, n, n, J))
c = torch.empty_like(a)
t(f"cdf97 double fwd {n}^2 J={J} out of place", lambda: dwt._fwd(dwt.CDF97_D, a, c, n * 8, 8, n, n, n, n, J, 0, 0, "fwd"))
t(f"cdf97 double inv {n}^2 J={J} out of place", lambda: dwt._inv(dwt.CDF97_D, c, a, n * 8, 8, n, n, n, n, J, 0, 0, "inv"))
dwt.set_option("fused_d", 0)
t(f"cdf97 double fwd {n}^2 J={J} in place, line passes", lambda: dwt.dwt_cdf97_2f_d(a, n * 8, 8, n, n, n, n, J))
dwt.set_option("fused_d", 1)
b = torch.rand((n, n), device="cuda", dtype=torch.float32)
t(f"cdf97 float  fwd {n}^2 J={J} in place (for scale)", lambda: dwt.dwt_cdf97_2f_s(b, n * 4, 4, n, n, n, n, J))
