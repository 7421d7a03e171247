#!/usr/bin/env python3
"""PCIe: H2D / D2H rates of 256 MiB from pinned and pageable host memory, alone and both
directions at once (two streams)."""
import time, torch
n = 64 << 20  # floats = 256 MiB
dev = torch.empty(n, device="cuda"); dev2 = torch.empty(n, device="cuda")
pin_a = torch.empty(n, pin_memory=True); pin_b = torch.empty(n, pin_memory=True)
pag_a = torch.empty(n); pag_b = torch.empty(n)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)
gb = n * 4 / 1e9
for name, a, b in (("pinned", pin_a, pin_b), ("pageable", pag_a, pag_b)):
    h2d = t(lambda: dev.copy_(a, non_blocking=True))
    d2h = t(lambda: b.copy_(dev2, non_blocking=True))
    def both():
        with torch.cuda.stream(s1): dev.copy_(a, non_blocking=True)
        with torch.cuda.stream(s2): b.copy_(dev2, non_blocking=True)
    bi = t(both)
    print(f"{name:9s}: H2D {h2d*1e3:6.2f} ms ({gb/h2d:5.1f} GB/s)  D2H {d2h*1e3:6.2f} ms ({gb/d2h:5.1f} GB/s)  both at once {bi*1e3:6.2f} ms ({2*gb/bi:5.1f} GB/s aggregate)", flush=True)
