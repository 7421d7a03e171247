#!/usr/bin/env python3
"""Soak of the interleaved in-place level (halo snapshot): random large shapes, pitches and level counts, 9/7 and 5/3,
forward and inverse, in place against out of place, bit for bit.  python scripts/archive/r04/il_inplace_soak.py [seconds] [seed]"""
import os, sys, time
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import libdwt_amd as dwt
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
dwt.dwt_util_init(); dwt.use_torch_stream()
t_end = time.time() + secs
n = bad = 0
while time.time() < t_end:
    h, w = int(rng.integers(1400, 6000)), int(rng.integers(1400, 6000))
    if rng.random() < 0.3:
        w = int(rng.integers(6, 24)) * 256 + int(rng.choice([0, 8, 9, 100, 255]))
    pitch = w + int(rng.integers(0, 9))
    J = int(rng.integers(1, 7)); wav = str(rng.choice(["cdf97_s", "cdf97_s", "cdf53_s"])); flav = int(rng.integers(0, 2))
    a = torch.rand((h, pitch), device="cuda")
    f = torch.full_like(a, 3)
    j = dwt.transform2d_interleaved(wav, 0, flav, a, f, pitch * 4, 4, w, h, None, None, J)
    x = a.clone()
    dwt.transform2d_interleaved(wav, 0, flav, x, x, pitch * 4, 4, w, h, None, None, J)
    ok = torch.equal(x[:, :w], f[:, :w]) and torch.equal(x[:, w:], a[:, w:])
    if flav == 0:
        r = torch.full_like(a, 4)
        dwt.transform2d_interleaved(wav, 1, 0, f, r, pitch * 4, 4, w, h, None, None, j)
        dwt.transform2d_interleaved(wav, 1, 0, x, x, pitch * 4, 4, w, h, None, None, j)
        ok = ok and torch.equal(x[:, :w], r[:, :w]) and torch.equal(x[:, w:], a[:, w:])
    n += 1
    if not ok:
        bad += 1
        print(f"MISMATCH: {wav} flavour {flav} {h}x{w} pitch {pitch} J={J}", flush=True)
torch.cuda.synchronize()
print(f"in-place soak: {n} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
