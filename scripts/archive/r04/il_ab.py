import os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
import libdwt_amd as dwt
n, J = 8192, 5
dwt.dwt_util_init(); dwt.use_torch_stream()
a = torch.rand((n, n), device="cuda"); b = torch.empty_like(a); c = a.clone()
def t(fn, reps=60):
    for _ in range(10): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for x, y in evs:
        x.record(); fn(); y.record()
    torch.cuda.synchronize()
    return statistics.median(x.elapsed_time(y) for x, y in evs) * 1e3
d = torch.empty_like(a)
for rnd in range(2):
    for v in (1, 0):
        dwt.set_option("il_exact_borders", v)
        f = t(lambda: dwt.transform2d_interleaved("cdf97_s", 0, 0, a, b, n*4, 4, n, n, None, None, J))
        fi = t(lambda: dwt.dwt_cdf97_2f_inplace_s(c, n*4, 4, n, n, n, n, J))
        i = t(lambda: dwt.transform2d_interleaved("cdf97_s", 1, 0, b, d, n*4, 4, n, n, None, None, J))
        ii = t(lambda: dwt.dwt_cdf97_2i_inplace_s(c, n*4, 4, n, n, n, n, J))
        print(f"il_exact_borders={v}: fwd out-of-place {f:.1f} us, in place {fi:.1f} us; inv out-of-place {i:.1f} us, in place {ii:.1f} us", flush=True)
dwt.set_option("il_exact_borders", 1)
dwt.transform2d_interleaved("cdf97_s", 0, 0, a, b, n*4, 4, n, n, None, None, J)
dwt.set_option("il_exact_borders", 0)
dwt.transform2d_interleaved("cdf97_s", 0, 0, a, d, n*4, 4, n, n, None, None, J)
torch.cuda.synchronize()
diff = (b - d).abs()
print("max |exact - fast| / max |coefficient|:", float(diff.max() / b.abs().max()), " differing samples:", int((diff > 0).sum()))
