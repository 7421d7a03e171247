#!/usr/bin/env python3
"""Timings of every entry of the path on one MI355X (for DESIGN.md): device-resident
out-of-place / in-place forward and inverse of float 9/7 at 8192^2 J=5, int 5/3 at
4096^2 J=3, the host-pointer (PCIe-inclusive) drop-in call, the batch of config 4 at
reduced count, and the 3-D path.  Run on the GPU box: python scripts/measure_entries.py"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import libdwt_amd as dwt

def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts), min(ts)

def main():
    dwt.dwt_util_init(); dwt.use_torch_stream()
    print("device", dwt.device_name())
    n, J = 8192, 5
    a = torch.rand((n, n), device="cuda"); b = torch.empty_like(a); c = a.clone()
    alg = sum(2*4*(n>>j)*(n>>j) for j in range(J))
    def rep(name, med, mn, samples, bytes_):
        print(f"{name:58s} median {med*1e6:9.1f} us  min {mn*1e6:9.1f} us  {samples/med/1e9:8.1f} Gsamples/s  {bytes_/med/1e9:7.0f} GB/s algorithmic")
    m = timeit(lambda: dwt.dwt_cdf97_2f_s2(a, b, n*4, 4, n, n, n, n, J)); rep("cdf97 fwd 8192^2 J=5 single image, out-of-place (_s2)", *m, n*n, alg)
    m = timeit(lambda: dwt.dwt_cdf97_2f_s(c, n*4, 4, n, n, n, n, J)); rep("cdf97 fwd 8192^2 J=5 single image, in-place", *m, n*n, alg)
    m = timeit(lambda: dwt.dwt_cdf97_2i_s2(b, c, n*4, 4, n, n, n, n, J)); rep("cdf97 inv 8192^2 J=5 single image, out-of-place (_s2)", *m, n*n, alg)
    m = timeit(lambda: dwt.dwt_cdf97_2i_s(c, n*4, 4, n, n, n, n, J)); rep("cdf97 inv 8192^2 J=5 single image, in-place", *m, n*n, alg)
    # interleaved (in-place lifting) layout
    m = timeit(lambda: dwt.transform2d_interleaved("cdf97_s", 0, 0, a, b, n*4, 4, n, n, None, None, J)); rep("cdf97 interleaved fwd 8192^2 J=5, out-of-place", *m, n*n, alg)
    m = timeit(lambda: dwt.dwt_cdf97_2f_inplace_s(c, n*4, 4, n, n, n, n, J)); rep("cdf97 interleaved fwd 8192^2 J=5, in-place", *m, n*n, alg)
    m = timeit(lambda: dwt.transform2d_interleaved("cdf97_s", 1, 0, b, c, n*4, 4, n, n, None, None, J)); rep("cdf97 interleaved inv 8192^2 J=5, out-of-place", *m, n*n, alg)
    m = timeit(lambda: dwt.dwt_cdf97_2i_inplace_s(c, n*4, 4, n, n, n, n, J)); rep("cdf97 interleaved inv 8192^2 J=5, in-place", *m, n*n, alg)
    dwt.set_option("il_exact_borders", 0)  # opt-in: no exact border strips (within 1e-5, not the reference's bits at the borders)
    m = timeit(lambda: dwt.transform2d_interleaved("cdf97_s", 0, 0, a, b, n*4, 4, n, n, None, None, J)); rep("  same, il_exact_borders=0: fwd out-of-place", *m, n*n, alg)
    m = timeit(lambda: dwt.transform2d_interleaved("cdf97_s", 1, 0, b, c, n*4, 4, n, n, None, None, J)); rep("  same, il_exact_borders=0: inv out-of-place", *m, n*n, alg)
    dwt.set_option("il_exact_borders", 1)
    nb = 8
    A = torch.rand((nb, n, n), device="cuda"); B = torch.empty_like(A)
    m = timeit(lambda: dwt.transform2d_batch("cdf97_s", 0, A, B, n*n*4, nb, n*4, n, n, J)); rep(f"cdf97 fwd 8192^2 J=5 batch of {nb}", *m, nb*n*n, nb*alg)
    m = timeit(lambda: dwt.transform2d_batch("cdf97_s", 1, B, A, n*n*4, nb, n*4, n, n, J)); rep(f"cdf97 inv 8192^2 J=5 batch of {nb}", *m, nb*n*n, nb*alg)
    del A, B
    # config 4 shape: 4096^2 J=5, 32 images per GPU
    n4, nb4 = 4096, 32
    A = torch.rand((nb4, n4, n4), device="cuda"); B = torch.empty_like(A)
    alg4 = sum(2*4*(n4>>j)*(n4>>j) for j in range(5))
    m = timeit(lambda: dwt.transform2d_batch("cdf97_s", 0, A, B, n4*n4*4, nb4, n4*4, n4, n4, 5)); rep(f"cdf97 fwd 4096^2 J=5 batch of {nb4} (config 4, per GPU)", *m, nb4*n4*n4, nb4*alg4)
    del A, B
    # config 3: int 5/3 4096^2 J=3
    n3 = 4096
    I = torch.randint(-32768, 32768, (n3, n3), device="cuda", dtype=torch.int32); O = torch.empty_like(I); I2 = I.clone()
    alg3 = sum(2*4*(n3>>j)*(n3>>j) for j in range(3))
    m = timeit(lambda: dwt._fwd(1, I, O, n3*4, 4, n3, n3, n3, n3, 3, 0, 0, "f")); rep("cdf53 int fwd 4096^2 J=3 single, out-of-place", *m, n3*n3, alg3)
    m = timeit(lambda: dwt.dwt_cdf53_2f_i(I2, n3*4, 4, n3, n3, n3, n3, 3)); rep("cdf53 int fwd 4096^2 J=3 single, in-place", *m, n3*n3, alg3)
    m = timeit(lambda: dwt._inv(1, O, I2, n3*4, 4, n3, n3, n3, n3, 3, 0, 0, "i")); rep("cdf53 int inv 4096^2 J=3 single, out-of-place", *m, n3*n3, alg3)
    IB = torch.randint(-32768, 32768, (16, n3, n3), device="cuda", dtype=torch.int32); OB = torch.empty_like(IB)
    m = timeit(lambda: dwt.transform2d_batch("cdf53_i", 0, IB, OB, n3*n3*4, 16, n3*4, n3, n3, 3)); rep("cdf53 int fwd 4096^2 J=3 batch of 16", *m, 16*n3*n3, 16*alg3)
    m = timeit(lambda: dwt.transform2d_batch("cdf53_i", 1, OB, IB, n3*n3*4, 16, n3*4, n3, n3, 3)); rep("cdf53 int inv 4096^2 J=3 batch of 16", *m, 16*n3*n3, 16*alg3)
    del IB, OB
    # host-pointer drop-in call (PCIe inclusive; ENTRIES_NO_HOST=1 skips it: counter passes)
    if os.environ.get("ENTRIES_NO_HOST"):
        h = None
    else:
      h = np.random.default_rng(0).random((n, n), dtype=np.float32)
      m = timeit(lambda: dwt.dwt_cdf97_2f_s(h, n*4, 4, n, n, n, n, J), reps=3, warm=1); rep("cdf97 fwd 8192^2 J=5 HOST pointer (H2D + kernels + D2H)", *m, n*n, alg)
      hs = np.random.default_rng(0).random((512, 512), dtype=np.float32)
      m = timeit(lambda: dwt.dwt_cdf97_2f_s(hs, 2048, 4, 512, 512, 512, 512, -1), reps=10, warm=2); rep("cdf97 fwd 512^2 full HOST pointer (examples/simple size)", *m, 512*512, sum(2*4*(512>>j)**2 for j in range(9)))
    # 3-D: in place (two passes per level) and out of place (one fused pass per level)
    for nn, lv in ((512, 3), (1024, 3)):
        try:
            V = torch.rand((nn, nn, nn), device="cuda"); O = torch.empty_like(V)
            vox = nn**3; algv = sum(8*((nn>>j)**3) for j in range(lv))
            m = timeit(lambda: dwt.transform3d_op(V, O, nn*4, nn*nn*4, nn, nn, nn, lv), reps=5, warm=2)
            rep(f"cdf97 3-D fwd {nn}^3 {lv} levels OUT OF PLACE (fused x+y+z levels)", *m, vox, algv)
            m = timeit(lambda: dwt.transform3d_op(V, O, nn*4, nn*nn*4, nn, nn, nn, 1), reps=5, warm=2)
            rep(f"cdf97 3-D fwd {nn}^3 1 level OUT OF PLACE (fused x+y+z)", *m, vox, 8*vox)
            del O
            m = timeit(lambda: dwt.transform3d(0, V, nn*4, nn*nn*4, nn, nn, nn, lv), reps=5, warm=2)
            rep(f"cdf97 3-D fwd {nn}^3 {lv} levels in place (two passes per level)", *m, vox, algv)
            m = timeit(lambda: dwt.transform3d(0, V, nn*4, nn*nn*4, nn, nn, nn, 1), reps=5, warm=2)
            rep(f"cdf97 3-D fwd {nn}^3 1 level in place", *m, vox, 8*vox)
            m = timeit(lambda: dwt.transform3d(1, V, nn*4, nn*nn*4, nn, nn, nn, 1), reps=5, warm=2)
            rep(f"cdf97 3-D inv {nn}^3 1 level in place", *m, vox, 8*vox)
            m = timeit(lambda: dwt.transform3d(1, V, nn*4, nn*nn*4, nn, nn, nn, lv), reps=5, warm=2)
            rep(f"cdf97 3-D inv {nn}^3 {lv} levels in place", *m, vox, algv)
            del V
        except Exception as e:
            print("3-D", nn, "failed:", e)

if __name__ == "__main__":
    main()
