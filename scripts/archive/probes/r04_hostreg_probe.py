#!/usr/bin/env python3
"""How long does pinning a caller's pageable image in place take (hipHostRegister), and how fast are copies from / to it
against the library's repack-through-a-pinned-buffer path?  (host-pointer calls: DESIGN s7)"""
import ctypes as C, time, os, sys
import numpy as np
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipDeviceSynchronize.argtypes = []
n = 8192
a = np.random.rand(n, n).astype(np.float32)
b = np.empty_like(a)
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), a.nbytes) == 0
for rep in range(3):
    t0 = time.perf_counter(); rc = hip.hipHostRegister(a.ctypes.data, a.nbytes, 0); t1 = time.perf_counter()
    assert rc == 0, rc
    hip.hipMemcpy(d, a.ctypes.data, a.nbytes, 1); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    hip.hipMemcpy(a.ctypes.data, d, a.nbytes, 2); hip.hipDeviceSynchronize(); t3 = time.perf_counter()
    hip.hipHostUnregister(a.ctypes.data); t4 = time.perf_counter()
    print(f"register {1e3*(t1-t0):.2f} ms, H2D {1e3*(t2-t1):.2f} ms ({a.nbytes/(t2-t1)/1e9:.1f} GB/s), D2H {1e3*(t3-t2):.2f} ms, unregister {1e3*(t4-t3):.2f} ms", flush=True)
# pageable copies straight through the runtime
for rep in range(3):
    t0 = time.perf_counter(); hip.hipMemcpy(d, b.ctypes.data, b.nbytes, 1); hip.hipDeviceSynchronize(); t1 = time.perf_counter()
    hip.hipMemcpy(b.ctypes.data, d, b.nbytes, 2); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    print(f"pageable: H2D {1e3*(t1-t0):.2f} ms ({b.nbytes/(t1-t0)/1e9:.1f} GB/s), D2H {1e3*(t2-t1):.2f} ms", flush=True)
# CPU copy rate, one thread and numpy
t0 = time.perf_counter(); np.copyto(b, a); t1 = time.perf_counter()
print(f"one-thread memcpy of the image: {1e3*(t1-t0):.2f} ms ({a.nbytes/(t1-t0)/1e9:.1f} GB/s)")
