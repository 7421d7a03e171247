#!/usr/bin/env python3
"""Interleaved entries: in place (level 0 over a halo snapshot) against out of place, bit for bit; where they differ."""
import os, sys
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
bad = 0
for (h, w, J) in [(8192, 8192, 5), (8192, 8192, 1), (4096, 4096, 3), (2048, 2048, 2), (3000, 5000, 3), (1024, 1024, 1), (6000, 2056, 2)]:
    a = torch.rand((h, w), device="cuda")
    for inv in (0, 1):
        src = a if inv == 0 else fwd_ref
        ref = torch.empty_like(a)
        dwt.transform2d_interleaved("cdf97_s", inv, 0, src, ref, w * 4, 4, w, h, None, None, J)
        if inv == 0:
            fwd_ref = ref.clone()
        for rep in range(3):
            x = src.clone()
            dwt.transform2d_interleaved("cdf97_s", inv, 0, x, x, w * 4, 4, w, h, None, None, J)
            torch.cuda.synchronize()
            d = (x != ref)
            n = int(d.sum())
            if n:
                bad += 1
                ys, xs = torch.nonzero(d, as_tuple=True)
                print(f"{h}x{w} J={J} inv={inv} rep={rep}: {n} differ; rows {int(ys.min())}..{int(ys.max())} cols {int(xs.min())}..{int(xs.max())}; "
                      f"row%128 hist top: {torch.bincount(ys % 128, minlength=128).topk(5)}; col%256 top: {torch.bincount(xs % 256, minlength=256).topk(5)}", flush=True)
                break
        else:
            print(f"{h}x{w} J={J} inv={inv}: in place == out of place", flush=True)
print("bad:", bad)
