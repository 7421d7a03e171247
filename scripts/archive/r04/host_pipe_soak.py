#!/usr/bin/env python3
"""Soak of the pipelined host-pointer calls: random shapes of 64 MiB and more, row pitches, level counts, wavelets, in place
and out of place, forward and inverse, against the plain path, bit for bit.  python scripts/archive/r04/host_pipe_soak.py [seconds] [seed]"""
import os, sys, time
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import numpy as np
import libdwt_amd as dwt
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
dwt.dwt_util_init()
t_end = time.time() + secs
n = bad = 0
while time.time() < t_end:
    h = int(rng.integers(2050, 12000)); w = max(64, int(np.ceil((16 << 20) / h)) + int(rng.integers(0, 3000)))
    pitch = w + int(rng.integers(0, 40))
    J = int(rng.choice([-1, 1, 2, 3, 5, 7])); wav = str(rng.choice(["cdf97_s", "cdf53_i", "cdf53_s", "cdf97_i"]))
    dt = np.int32 if wav.endswith("_i") else np.float32
    base = (rng.integers(-20000, 20000, size=(h, pitch)).astype(np.int32) if dt == np.int32 else rng.random((h, pitch), dtype=np.float32))
    res = []
    for pipe in (0, 1):
        dwt.set_option("host_pipeline", pipe)
        a = base.copy()
        j = dwt._fwd(dwt.WAVELET_ID[wav], a, a, pitch * 4, 4, w, h, w, h, J, 0, 0, "f")
        b = np.full_like(base, 3)
        dwt._fwd(dwt.WAVELET_ID[wav], base, b, pitch * 4, 4, w, h, w, h, J, 0, 0, "f")
        ok = np.array_equal(a[:, :w].view(np.uint32), b[:, :w].view(np.uint32)) and np.array_equal(a[:, w:], base[:, w:]) and np.all(b[:, w:] == 3)
        fwd = a[:, :w].copy()
        c = np.full_like(base, 4)
        dwt._inv(dwt.WAVELET_ID[wav], a, c, pitch * 4, 4, w, h, w, h, j, 0, 0, "i")
        dwt._inv(dwt.WAVELET_ID[wav], a, a, pitch * 4, 4, w, h, w, h, j, 0, 0, "i")
        ok = ok and np.array_equal(a[:, :w].view(np.uint32), c[:, :w].view(np.uint32)) and np.array_equal(a[:, w:], base[:, w:])
        res.append((j, fwd, a[:, :w].copy(), ok))
    dwt.set_option("host_pipeline", 1)
    ok = res[0][3] and res[1][3] and res[0][0] == res[1][0] and np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32)) and \
        np.array_equal(res[0][2].view(np.uint32), res[1][2].view(np.uint32))
    n += 1
    if not ok:
        bad += 1
        print(f"MISMATCH: {wav} {h}x{w} pitch {pitch} J={J}", flush=True)
print(f"host pipeline soak: {n} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
