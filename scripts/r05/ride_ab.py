#!/usr/bin/env python3
"""In-place entries on one 8192^2 image (dwt_cdf97_2f_s / dwt_cdf97_2i_s, device pointers): the staged subbands' copy as a
launch of its own (ride_copy = 0) against riding along with the deeper levels' launches (1), alternated in one process;
ride_mib = MiB of copy per small level.  Same bits either way (checked)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, J, nb = int(os.environ.get("SIZE", 8192)), int(os.environ.get("LEVELS", 5)), 12
src = torch.rand((nb, n, n), device="cuda"); work = torch.empty_like(src)

def timed(fn, reps=60):
    for i in range(8): fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(i); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[0]

fwd = lambda i: dwt.dwt_cdf97_2f_s(work[i % nb], n * 4, 4, n, n, n, n, J)
inv = lambda i: dwt.dwt_cdf97_2i_s(work[i % nb], n * 4, 4, n, n, n, n, J)
# bits
res = {}
for ride in (0, 1):
    dwt.set_option("ride_copy", ride)
    work.copy_(src)
    for k in range(2): fwd(k)
    torch.cuda.synchronize(); f = work[:2].clone()
    for k in range(2): inv(k)
    torch.cuda.synchronize(); res[ride] = (f, work[:2].clone())
same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(res[0], res[1]))
print("same bits with and without the ride-along copy:", same, " round trip error %.2e" % float((res[1][1] - src[:2]).abs().max()), flush=True)
for rnd in range(2):
    for ride, mib in ((0, 48), (1, 8), (1, 16), (1, 24), (1, 32), (1, 48)):
        dwt.set_option("ride_copy", ride); dwt.set_option("ride_mib", mib)
        work.copy_(src)
        f = timed(fwd)
        dwt.transform2d_batch("cdf97_s", 0, src, work, n * n * 4, nb, n * 4, n, n, J)
        b = timed(inv)
        print(f"ride_copy {ride} ride_mib {mib:3d}: in-place forward median {f[0]:6.1f} min {f[1]:6.1f} us   inverse median {b[0]:6.1f} min {b[1]:6.1f} us", flush=True)
dwt.set_option("ride_copy", 1); dwt.set_option("ride_mib", 48)
