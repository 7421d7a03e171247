#!/usr/bin/env python3
"""In-place single-image entries, A/B of an option inside one process (HIP-event medians over 100 calls,
each call on another image of a resident batch): python scripts/archive/r04/inplace_ab.py opt=v0,v1 [inverse]"""
import os, sys, statistics
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
name, vals = sys.argv[1].split("=")
vals = [int(v) for v in vals.split(",")]
inverse = len(sys.argv) > 2
n, J, nb = 8192, 5, 16
dwt.dwt_util_init(); dwt.use_torch_stream()
work = torch.rand((nb, n, n), device="cuda")
def run(i):
    k = i % nb
    (dwt.dwt_cdf97_2i_s if inverse else dwt.dwt_cdf97_2f_s)(work[k], n * 4, 4, n, n, n, n, J)
for rnd in range(3):
    for v in vals:
        dwt.set_option(name, v)
        for i in range(20): run(i)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
        for i, (a, b) in enumerate(evs):
            a.record(); run(i); b.record()
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in evs]
        print(f"{name}={v}: median {statistics.median(ms)*1e3:7.1f} us  min {min(ms)*1e3:7.1f} us", flush=True)
