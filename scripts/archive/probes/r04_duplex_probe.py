#!/usr/bin/env python3
"""PCIe both ways at once from pinned memory: what a host-pointer call could overlap at best."""
import time, torch
n = 256 << 20
h_up = torch.empty(n, dtype=torch.uint8).pin_memory(); h_dn = torch.empty(n, dtype=torch.uint8).pin_memory()
d_up = torch.empty(n, dtype=torch.uint8, device="cuda"); d_dn = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(up, dn, pieces=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k = n // pieces
    for i in range(pieces):
        if up:
            with torch.cuda.stream(s1): d_up[i*k:(i+1)*k].copy_(h_up[i*k:(i+1)*k], non_blocking=True)
        if dn:
            with torch.cuda.stream(s2): h_dn[i*k:(i+1)*k].copy_(d_dn[i*k:(i+1)*k], non_blocking=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for _ in range(2):
    print(f"256 MiB up only {run(1,0):.2f} ms, down only {run(0,1):.2f} ms, both at once {run(1,1):.2f} ms, both in 8 pieces {run(1,1,8):.2f} ms, both in 32 pieces {run(1,1,32):.2f} ms")
