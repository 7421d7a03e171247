#!/usr/bin/env python3
"""Fast / slow placement, round 4: map of the coarse physical regions.  One big allocation; a SMALL batch
(4 images: 1 GiB) whose destination / LL scratch are moved through the pool in 1 GiB steps."""
import os, sys
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, nb = 8192, int(os.environ.get("IMAGES", 4))
img = n * n * 4
GB = 1 << 30
dwt.dwt_util_init(); dwt.use_torch_stream()
free_b, total_b = torch.cuda.mem_get_info()
pool_gb = (free_b - 4 * GB) // GB
pool = torch.empty(pool_gb * GB, dtype=torch.uint8, device="cuda")
print(f"pool {pool_gb} GiB at {pool.data_ptr():#x}; {nb} images", flush=True)
batch_b = nb * img
b0, b1 = nb * (n // 2) ** 2 * 4 + 4096, nb * (n // 4) ** 2 * 4 + 4096
def view(off_gb, nbytes):
    o = int(off_gb * GB)
    return pool[o:o + nbytes]
def rate(src, dst, J, reps=6):
    for _ in range(2): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n * 4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(2)
    for _ in range(reps): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n * 4, n, n, J)
    torch.cuda.synchronize(); ms, cnt = dwt.prof_read_levels(4); dwt.prof_enable(False)
    return 2 * 4 * n * n * nb / ms[0] / 1e6
def ws(P):
    w0 = view(P, b0); w1 = view(P + 0.5, b1)
    assert dwt.lib.dwt_hip_set_workspace(w0.data_ptr(), b0, w1.data_ptr(), b1) == 0
src = view(0, batch_b); src.view(torch.float32).uniform_()
step = float(os.environ.get("STEP_GB", 1))
for D in (2, 100):
    dst = view(D, batch_b)
    print(f"-- source at 0, destination at {D}, scratch at P = 4, 4+{step}, ... (level-0 GB/s / 100)")
    row = []
    P = 4.0
    while P + 1 < pool_gb:
        if abs(P - D) >= 1.0:
            ws(P); row.append(f"{rate(src, dst, 2) / 100:.0f}")
        else:
            row.append("--")
        P += step
    print(" ".join(row), flush=True)
ws(4)
print("-- source at 0, scratch at 4, destination at D = 6, 7, ...")
row = []
D = 6.0
while D + 1 < pool_gb:
    row.append(f"{rate(src, view(D, batch_b), 2) / 100:.0f}")
    D += step
print(" ".join(row), flush=True)
print("-- J=1 (no scratch): source at 0, destination at D = 2, 3, ...")
row = []
D = 2.0
while D + 1 < pool_gb:
    row.append(f"{rate(src, view(D, batch_b), 1) / 100:.0f}")
    D += step
print(" ".join(row), flush=True)
dwt.lib.dwt_hip_set_workspace(None, 0, None, 0)
