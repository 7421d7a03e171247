#!/usr/bin/env python3
"""Per-level kernel times of the batched forward (or inverse) transform on the GPU box.
    IMAGES=4 [SIZE=8192 LEVELS=5 INVERSE=0 WAVELET=cdf97_s] python scripts/levels.py ["opt=val,..."]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt

n = int(os.environ.get("SIZE", 8192)); J = int(os.environ.get("LEVELS", 5)); nb = int(os.environ.get("IMAGES", 4))
inv = int(os.environ.get("INVERSE", 0))
wav = os.environ.get("WAVELET", "cdf97_s")
dwt.dwt_util_init(); dwt.use_torch_stream()
for v in sys.argv[1:] or [""]:
    for kv in [x for x in v.split(",") if x]:
        k, val = kv.split("="); dwt.set_option(k, int(val))
    src = torch.randint(-32768, 32768, (nb, n, n), device="cuda", dtype=torch.int32) if wav.endswith("_i") else torch.rand((nb, n, n), device="cuda")
    dst = torch.empty_like(src)
    if inv:
        dwt.transform2d_batch(wav, 0, src, dst, n*n*4, nb, n*4, n, n, J); src, dst = dst, src
    for _ in range(3): dwt.transform2d_batch(wav, inv, src, dst, n*n*4, nb, n*4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(2)
    t0 = time.perf_counter()
    for _ in range(10): dwt.transform2d_batch(wav, inv, src, dst, n*n*4, nb, n*4, n, n, J)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10
    ms, cnt = dwt.prof_read_levels(J); dwt.prof_enable(0)
    parts = "  ".join(f"L{j}: {ms[j]*1e3:7.1f}us ({2*4*(n>>j)**2*nb/ms[j]/1e6:6.0f} GB/s)" for j in range(J))
    print(f"images={nb} {v:30s} step {el*1e3:.4f} ms (with events)  sum {sum(ms)*1e3:.1f}us | {parts}")
    del src, dst
