#!/usr/bin/env python3
"""Round 4: is the two-stream write probe a proxy for the sweep?  One pool, 16-image batches (4 GiB buffers,
inside one 16 GiB region).  Per 4 GiB step P: probe(dst region, P) | level-0 rate with dst fixed and the LL
scratch at P | probe(src region, P) | level-0 rate (J=1) with src fixed and the destination at P."""
import os, sys
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, nb = 8192, 16
img = n * n * 4
GB = 1 << 30
dwt.dwt_util_init(); dwt.use_torch_stream()
L = dwt.lib
free_b, total_b = torch.cuda.mem_get_info()
pool_gb = (free_b - 4 * GB) // GB
pool = torch.empty(pool_gb * GB, dtype=torch.uint8, device="cuda")
base = pool.data_ptr()
batch_b = nb * img
b0, b1 = nb * (n // 2) ** 2 * 4 + 4096, nb * (n // 4) ** 2 * 4 + 4096
def view(off_gb, nbytes):
    o = int(off_gb * GB); return pool[o:o + nbytes]
def rate(src, dst, J, reps=4):
    for _ in range(2): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n * 4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(2)
    for _ in range(reps): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n * 4, n, n, J)
    torch.cuda.synchronize(); ms, cnt = dwt.prof_read_levels(4); dwt.prof_enable(False)
    return 2 * 4 * n * n * nb / ms[0] / 1e6
def ws(P):
    w0 = view(P, b0); w1 = view(P + 1.25, b1)
    assert L.dwt_hip_set_workspace(w0.data_ptr(), b0, w1.data_ptr(), b1) == 0
PB = 256 << 20
def cprobe(a_gb, b_gb, nbytes=256 << 20):
    return 2 * nbytes / L.dwt_hip_probe_copy_us(base + int(a_gb * GB), base + int(b_gb * GB), nbytes) / 1e3
def probe(a_gb, b_gb):
    return 2 * PB / L.dwt_hip_probe_pair_us(base + int(a_gb * GB), base + int(b_gb * GB), PB) / 1e3
S, D = 0, 4   # source at 0..4, destination at 4..8 (same 16 GiB region as the source, as a plain allocation would be)
src = view(S, batch_b); src.view(torch.float32).uniform_()
dst = view(D, batch_b)
print("P | copy probe 256 MiB src region -> P, P -> src region, 1 GiB src region -> P || P | probe(dst,P) | level0 J=2 (src 0, dst 4, scratch P) | probe(src,P) | level0 J=1 (src 0, dst P) | level0 J=2 (src 0, dst P, scratch 8)")
for P in range(8, pool_gb - 5, 4):
    ws(P)
    r2 = rate(src, dst, 2)
    ws(8)
    r1 = rate(src, view(P, batch_b), 1) if P >= 8 else 0
    r2d = rate(src, view(P, batch_b), 2) if P >= 12 else 0
    print(f"{P:4d} | {cprobe(S + 1, P):5.0f} {cprobe(P, S + 1):5.0f} {cprobe(S + 1, P, 1 << 30):5.0f} || {P:4d} | {probe(D, P):5.0f} | {r2:5.0f} | {probe(S + 1, P):5.0f} | {r1:5.0f} | {r2d:5.0f}", flush=True)
L.dwt_hip_set_workspace(None, 0, None, 0)
