#!/usr/bin/env python3
"""Host-pointer calls, 8192^2 float 9/7, 5 levels, forward and inverse: pipelined under the transfers against upload / transform / download."""
import os, sys, time
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import numpy as np
import libdwt_amd as dwt
dwt.dwt_util_init()
n, J = 8192, 5
rng = np.random.default_rng(1)
img = rng.random((n, n), dtype=np.float32)
ref = None
for rnd in range(3):
    for pipe in (0, 1):
        dwt.set_option("host_pipeline", pipe)
        a = img.copy(); b = np.empty_like(a)
        ts = []
        for rep in range(4):
            a[:] = img
            t0 = time.perf_counter(); dwt.dwt_cdf97_2f_s(a, n * 4, 4, n, n, n, n, J); t1 = time.perf_counter()
            ts.append(t1 - t0)
        t2 = []
        for rep in range(4):
            t0 = time.perf_counter(); dwt.dwt_cdf97_2f_s2(img, b, n * 4, 4, n, n, n, n, J); t1 = time.perf_counter()
            t2.append(t1 - t0)
        if ref is None:
            ref = b.copy()
        ok = np.array_equal(a, ref) and np.array_equal(b, ref)
        ti = []
        for rep in range(4):
            a[:] = ref
            t0 = time.perf_counter(); dwt.dwt_cdf97_2i_s(a, n * 4, 4, n, n, n, n, J); t1 = time.perf_counter()
            ti.append(t1 - t0)
        print(f"host_pipeline={pipe}: inverse in place {min(ti)*1e3:.2f} ms, round trip error {np.abs(a - img).max():.2e}; forward results as the plain path's: {ok}; in place {min(ts)*1e3:.2f} ms (median {sorted(ts)[len(ts)//2]*1e3:.2f}), out of place {min(t2)*1e3:.2f} ms", flush=True)
