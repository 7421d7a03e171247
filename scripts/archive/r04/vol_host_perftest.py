import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import libdwt_amd as dwt
dwt.dwt_util_init()
for n in (256, 512):
    for opt in (1, 0):
        err, secs = dwt.volume_perftest_fwd97op_s(n, opt, 0, 3)
        print(f"volume_perftest_fwd97op_s({n}, opt_stride={opt}): {err} errors, {secs*1e9:.4f} ns per voxel", flush=True)
