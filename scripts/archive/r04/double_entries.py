import os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
def t(fn, reps=20):
    for _ in range(4): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for x, y in evs:
        x.record(); fn(); y.record()
    torch.cuda.synchronize()
    return statistics.median(x.elapsed_time(y) for x, y in evs) * 1e3
n, J = 8192, 5
for wav, wid in (("cdf97_d", 3), ("cdf53_d", 4)):
    a = torch.rand((n, n), device="cuda", dtype=torch.float64); b = torch.empty_like(a); c = torch.empty_like(a)
    f = t(lambda: dwt._fwd(wid, a, b, n*8, 8, n, n, n, n, J, 0, 0, "f"))
    i = t(lambda: dwt._inv(wid, b, c, n*8, 8, n, n, n, n, J, 0, 0, "i"))
    alg = sum(2*8*(n>>j)**2 for j in range(J))
    print(f"{wav} 8192^2 J=5 single: fwd {f:.1f} us ({alg/f/1e3:.0f} GB/s), inv {i:.1f} us ({alg/i/1e3:.0f} GB/s); round trip err {(c-a).abs().max().item():.2e}", flush=True)
    A = torch.rand((8, n, n), device="cuda", dtype=torch.float64); B = torch.empty_like(A)
    fb = t(lambda: dwt.transform2d_batch(wav, 0, A, B, n*n*8, 8, n*8, n, n, J), reps=10)
    print(f"{wav} batch of 8: fwd {fb:.1f} us ({8*alg/fb/1e3:.0f} GB/s)", flush=True)
    del A, B
