#!/usr/bin/env python3
"""Copy-bandwidth ceiling vs footprint (torch copy_ = the runtime's blit kernel)."""
import torch, time
for mib in (8, 16, 32, 64, 128, 256, 1024, 4096):
    n = mib * 1024 * 1024 // 4
    a = torch.rand(n, device="cuda"); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 200 if mib <= 128 else 10
    for _ in range(reps): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"copy {mib:5d} MiB -> {mib:5d} MiB: {ms*1e3:8.1f} us  {2*n*4/ms/1e9:6.2f} TB/s", flush=True)
    del a, b
