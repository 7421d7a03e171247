#!/usr/bin/env python3
"""2-D copies between a host image with an odd byte pitch (libdwt's 'optimal' strides are odd) and the device: pageable,
registered in place, against the library's repack-through-pinned path."""
import ctypes as C, time
import numpy as np
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy2D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
hip.hipDeviceSynchronize.argtypes = []
w, h = 2048, 32768   # 256 MiB of samples
d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), w * 4 * h) == 0
for pitch in (w * 4, w * 4 + 64, w * 4 + 4, w * 4 + 13):
    buf = np.zeros(pitch * h + 64, np.uint8)
    p = buf.ctypes.data + (-buf.ctypes.data) % 16
    for reg in (0, 1):
        if reg: assert hip.hipHostRegister(p, pitch * h, 0) == 0
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); rc = hip.hipMemcpy2D(d, w * 4, p, pitch, w * 4, h, 1); hip.hipDeviceSynchronize(); t1 = time.perf_counter()
            rc2 = hip.hipMemcpy2D(p, pitch, d, w * 4, w * 4, h, 2); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1, rc, rc2))
        if reg: hip.hipHostUnregister(p)
        up, dn, rc, rc2 = min(ts)
        print(f"pitch {pitch} B {'registered' if reg else 'pageable  '}: H2D {up*1e3:7.2f} ms ({w*4*h/up/1e9:5.1f} GB/s)  D2H {min(t[1] for t in ts)*1e3:7.2f} ms  rc {rc} {rc2}", flush=True)
