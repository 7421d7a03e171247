#!/usr/bin/env python3
"""Round 3: the one-pass in-place 3-D levels (dwt_hip_transform3d) at full size.
 * bits: in-place forward == out-of-place forward (dwt_hip_transform3d_op), in-place inverse (one
   pass) == in-place inverse (two passes through scratch), 1024^3 and a ragged volume;
 * time: forward / inverse, 1 and 3 levels, for vol_inplace_fused = 1 (one pass over a halo
   snapshot), 2 (round 2: fused out of place + copy back, forward only), 0 (two passes).
python scripts/archive/r03/r03_vol_ip.py [n]"""
import os, sys, time, statistics
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dwt.dwt_util_init(); dwt.use_torch_stream()

def run(inv, a, lv):
    nz, ny, nx = a.shape
    dwt.transform3d(inv, a, nx * 4, nx * ny * 4, nx, ny, nz, lv)

for shape, lv in [((n, n, n), 3), ((n, n, n), 1), ((301, 1000, 1111), 2)]:
    nz, ny, nx = shape
    a = torch.rand(shape, device="cuda")
    ref = torch.empty_like(a)
    dwt.transform3d_op(a, ref, nx * 4, nx * ny * 4, nx, ny, nz, lv)
    f = a.clone()
    run(0, f, lv)
    torch.cuda.synchronize()
    same_f = torch.equal(f, ref)
    g = ref.clone()
    run(1, f, lv)                      # one pass per level
    dwt.set_option("vol_inplace_fused", 0)
    run(1, g, lv)                      # two passes per level
    dwt.set_option("vol_inplace_fused", 1)
    torch.cuda.synchronize()
    same_i = torch.equal(f, g)
    err = (f - a).abs().max().item()
    print(f"{shape} J={lv}: in-place forward == out-of-place: {same_f}; one-pass inverse == two-pass: {same_i}; round trip {err:.2e}", flush=True)
    del a, ref, f, g

a = torch.rand((n, n, n), device="cuda")
for mode in (1, 0):
    dwt.set_option("vol_inplace_fused", mode)
    for levels in (1, 3):
        for inverse in (0, 1):
            fn = lambda: run(inverse, a, levels)
            for _ in range(2): fn()
            torch.cuda.synchronize(); ts = []
            for _ in range(7):
                t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            t = statistics.median(ts)
            alg = sum(8 * ((n >> j) ** 3) for j in range(levels))
            print(f"vol_inplace_fused={mode} {'inverse' if inverse else 'forward'} {n}^3 {levels} level(s): {t*1e3:8.3f} ms  "
                  f"{n**3/t/1e9:7.1f} Gvoxel/s  {alg/t/8e12:.3f} of 8 TB/s", flush=True)
dwt.set_option("vol_inplace_fused", 1)
