#!/usr/bin/env python3
"""Bounded soak of the placed-batch x sharded-entry sequence of tests/test_hip_multi.py (both wavelets, 25 rounds, one
process): on a mismatch prints which images differ, whether the source on the device is intact and whether the result
arrives late (a second read after a device-wide synchronisation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import libdwt_amd as dwt
from oraclelib import Oracle
dwt.dwt_util_init()
orc = Oracle()
L = dwt.lib
nb, h, w, J = 7, 260, 520, 3
bad = 0
WAVS = [w for w in (("cdf97_s", "cdf97_2f_s", np.float32), ("cdf53_i", "cdf53_2f_i", np.int32)) if w[0] in os.environ.get("WAVS", "cdf97_s,cdf53_i")]
FINISH = int(os.environ.get("FINISH", 1))
for rnd in range(int(os.environ.get("ROUNDS", 25))):
    for wname, ff, dt in WAVS:
        rng = np.random.default_rng(rnd)
        imgs = rng.integers(-32768, 32768, size=(nb, h, w), dtype=np.int32) if dt == np.int32 else rng.random((nb, h, w), dtype=np.float32) * 2 - 1
        want = imgs.copy()
        for k in range(nb):
            orc.fwd(ff, want[k], J)
        dwt.set_option("place_min_mib", 0); dwt.set_option("place_tries", 2); dwt.set_option("place_max_gib", 24)
        if FINISH:
            dwt.dwt_util_finish()
        src, dst = dwt.alloc_batch(wname, nb, w, h, J)
        rep = dwt.alloc_batch_report()
        print(f"round {rnd} {wname}: src {src:#x} dst {dst:#x} note '{dwt.alloc_batch_note()}' arena {rep.get('arena_GiB')} GiB dst_at {rep.get('dst_at')} ll_at {rep.get('ll_at')}", flush=True)
        assert L.dwt_hip_memcpy_h2d(src, imgs.ctypes.data, imgs.nbytes) == 0
        got = np.empty_like(imgs); zeros = np.zeros_like(imgs)
        for devices in eval(os.environ.get("DEVLISTS", "([0, 0, 0], [0], [0, 0, 0, 0, 0])")):
            assert L.dwt_hip_memcpy_h2d(dst, zeros.ctypes.data, imgs.nbytes) == 0
            assert dwt.transform2d_batch_sharded(wname, 0, src, dst, h * w * 4, nb, w * 4, w, h, J, devices) == J
            assert L.dwt_hip_memcpy_d2h(got.ctypes.data, dst, got.nbytes) == 0
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                bad += 1
                diff = [k for k in range(nb) if not np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32))]
                zero = [k for k in range(nb) if not got[k].any()]
                back = np.empty_like(imgs)
                L.dwt_hip_memcpy_d2h(back.ctypes.data, src, back.nbytes)
                dwt.sync()
                late = np.empty_like(imgs)
                L.dwt_hip_memcpy_d2h(late.ctypes.data, dst, late.nbytes)
                print(f"round {rnd} {wname} devices {devices}: images differing {diff}, all-zero {zero}, source intact {np.array_equal(back, imgs)}, "
                      f"late read equal {np.array_equal(late.view(np.uint32), want.view(np.uint32))}", flush=True)
        L.dwt_hip_free(src); L.dwt_hip_free(dst)
print("mismatches:", bad)
