#!/usr/bin/env python3
"""Config 5 on plain first allocations of the two volumes: does it matter where the LIBRARY's scratch pools land?
The pools are (re)allocated behind spacers of different sizes."""
import os, sys, statistics
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, J = 1024, 3
dwt.dwt_util_init(); dwt.use_torch_stream()
dwt.set_option("place_tries", 1)
src = torch.rand((n, n, n), device="cuda"); dst = torch.empty_like(src)
fn = lambda: dwt.transform3d_op(src, dst, n * 4, n * n * 4, n, n, n, J)
def t():
    for _ in range(4): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in evs)
print(f"pools right behind the volumes: {t():.4f} ms", flush=True)
for gib in (4, 8, 16, 24, 32, 48, 64, 96):
    dwt.dwt_util_finish(); dwt.dwt_util_init(); dwt.use_torch_stream(); dwt.set_option("place_tries", 1)
    torch.cuda.empty_cache()
    spacer = torch.empty(gib << 30, dtype=torch.uint8, device="cuda")
    fn(); torch.cuda.synchronize()
    del spacer; torch.cuda.empty_cache()
    print(f"pools behind a spacer of {gib:3d} GiB: {t():.4f} ms", flush=True)
