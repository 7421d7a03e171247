import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
mode = sys.argv[1]
if mode != "notorch":
    import torch
import libdwt_amd as dwt
dwt.dwt_util_init()
if mode == "torchstream":
    dwt.use_torch_stream()
n, J = 8192, 5
img = np.random.default_rng(1).random((n, n), dtype=np.float32)
for pipe in (1, 0, 1):
    dwt.set_option("host_pipeline", pipe)
    a = img.copy(); ts = []
    for rep in range(4):
        a[:] = img
        t0 = time.perf_counter(); dwt.dwt_cdf97_2f_s(a, n * 4, 4, n, n, n, n, J); ts.append(time.perf_counter() - t0)
    print(mode, "pipe", pipe, [f"{t*1e3:.2f}" for t in ts], flush=True)
