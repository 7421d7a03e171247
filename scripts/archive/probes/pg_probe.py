import os, sys, datetime, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
import torch, torch.distributed as dist
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
t0 = time.time()
plane, control = bench.open_group(torch, dist, dev, 0, False, 120)
print("control plane:", control, "on_device", plane.on_device, f"{time.time()-t0:.1f}s")
plane.barrier(); print("barrier ok"); print("max", plane.max(1.5))
src = torch.rand((4, 64, 64), device=dev)
print("split", bench.batch_split_times(torch, plane, src, 4, 64, 0, 1, dev))
dist.destroy_process_group(); print("done")
