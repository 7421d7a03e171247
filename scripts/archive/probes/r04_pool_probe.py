#!/usr/bin/env python3
"""Fast / slow placement, round 4: ONE big allocation (most of the card); source, destination and the
LL scratch placed at chosen offsets inside it.  Does the level-0 rate depend on WHERE in physical
memory the streams lie relative to each other (coarse regions), rather than on fine address bits?"""
import os, sys
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n, nb = 8192, 64
img = n * n * 4
GB = 1 << 30
dwt.dwt_util_init(); dwt.use_torch_stream()
free_b, total_b = torch.cuda.mem_get_info()
pool_gb = int(os.environ.get("POOL_GB", (free_b - 6 * GB) // GB))
pool = torch.empty(pool_gb * GB, dtype=torch.uint8, device="cuda")
print(f"free {free_b/GB:.1f} GiB of {total_b/GB:.1f}; pool {pool_gb} GiB at {pool.data_ptr():#x}", flush=True)
batch_b = nb * img  # 16 GiB
def view(off_gb, nbytes):
    o = int(off_gb * GB)
    return pool[o:o + nbytes]
def rate(src, dst, J, reps=4):
    for _ in range(2): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n * 4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(2)
    for _ in range(reps): dwt.transform2d_batch("cdf97_s", 0, src, dst, img, nb, n * 4, n, n, J)
    torch.cuda.synchronize(); ms, cnt = dwt.prof_read_levels(4); dwt.prof_enable(False)
    return 2 * 4 * n * n * nb / ms[0] / 1e6, (2 * 4 * (n // 2) ** 2 * nb / ms[1] / 1e6 if J > 1 else 0.0)
step = int(os.environ.get("STEP_GB", 8))
print("-- J=1 (no scratch): source at S, destination at D (GiB offsets in the pool)")
for S in (0, pool_gb // 2):
    src = view(S, batch_b); src.view(torch.float32).uniform_()
    row = []
    for D in range(0, pool_gb - 16, step):
        if abs(D - S) < 16: continue
        r, _ = rate(src, view(D, batch_b), 1)
        row.append(f"D={D}:{r:.0f}")
    print(f"S={S}: " + "  ".join(row), flush=True)
print("-- J=2: source at 0, destination at 16, both LL bands (4+1 GiB) at P")
src = view(0, batch_b); src.view(torch.float32).uniform_()
dst = view(16, batch_b)
b0, b1 = nb * (n // 2) ** 2 * 4 + 4096, nb * (n // 4) ** 2 * 4 + 4096
row = []
for P in range(32, pool_gb - 6, step):
    w0 = view(P, b0); w1 = view(P + 4.5, b1)
    assert dwt.lib.dwt_hip_set_workspace(w0.data_ptr(), b0, w1.data_ptr(), b1) == 0
    r0, r1 = rate(src, dst, 2)
    row.append(f"P={P}:{r0:.0f}/{r1:.0f}")
print("  ".join(row), flush=True)
print("-- J=2: LL bands at 40, source at 0, destination at D")
w0 = view(40, b0); w1 = view(44.5, b1)
dwt.lib.dwt_hip_set_workspace(w0.data_ptr(), b0, w1.data_ptr(), b1)
row = []
for D in range(48, pool_gb - 16, step):
    r0, r1 = rate(src, view(D, batch_b), 2)
    row.append(f"D={D}:{r0:.0f}/{r1:.0f}")
print("  ".join(row), flush=True)
print("-- J=2: destination at 16, LL bands at 40, source at S")
row = []
for S in range(48, pool_gb - 16, step):
    s = view(S, batch_b); s.view(torch.float32).uniform_()
    r0, r1 = rate(s, dst, 2)
    row.append(f"S={S}:{r0:.0f}/{r1:.0f}")
print("  ".join(row), flush=True)
dwt.lib.dwt_hip_set_workspace(None, 0, None, 0)
