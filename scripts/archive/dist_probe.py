#!/usr/bin/env python3
"""Does an RCCL communicator in the process slow the transform kernels?  Run under torchrun."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import libdwt_amd as dwt
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dwt.dwt_util_init(); dwt.use_torch_stream()
n, nb, J = 8192, 8, 5
a = torch.rand((nb, n, n), device="cuda"); b = torch.empty_like(a)
def measure(tag):
    for _ in range(5): dwt.transform2d_batch("cdf97_s", 0, a, b, n*n*4, nb, n*4, n, n, J)
    torch.cuda.synchronize(); dwt.prof_enable(True); t0 = time.perf_counter()
    for _ in range(20): dwt.transform2d_batch("cdf97_s", 0, a, b, n*n*4, nb, n*4, n, n, J)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20
    ms, cnt = dwt.prof_read(); dwt.prof_enable(False)
    print(f"{tag:42s} step {el*1e3:.4f} ms  L0 {ms/cnt:.4f} ms", flush=True)
measure("before init_process_group")
dist.init_process_group("nccl")
measure("after init_process_group (no collective yet)")
dist.barrier(); torch.cuda.synchronize()
measure("after the first barrier (communicator up)")
measure("again")
dist.destroy_process_group()
measure("after destroy_process_group")

# what does the barrier right before a short timed region cost?
dist.init_process_group("nccl")
gl = dist.new_group(backend="gloo")
dist.barrier(); torch.cuda.synchronize()
def timed(tag, pre):
    res = []
    for rep in range(4):
        for _ in range(5): dwt.transform2d_batch("cdf97_s", 0, a, b, n*n*4, nb, n*4, n, n, J)
        torch.cuda.synchronize(); pre(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): dwt.transform2d_batch("cdf97_s", 0, a, b, n*n*4, nb, n*4, n, n, J)
        torch.cuda.synchronize(); pre(); torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 5)
    print(f"{tag:42s} 5-step regions: " + " ".join(f"{r*1e3:.4f}" for r in res) + " ms/step", flush=True)
timed("bracket: synchronize only", lambda: None)
timed("bracket: RCCL barrier", lambda: dist.barrier())
timed("bracket: gloo barrier", lambda: dist.barrier(group=gl))
timed("bracket: RCCL barrier", lambda: dist.barrier())
dist.destroy_process_group()
