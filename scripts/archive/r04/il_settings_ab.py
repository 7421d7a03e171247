import os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
import libdwt_amd as dwt
n, J = 8192, 5
dwt.dwt_util_init(); dwt.use_torch_stream()
a = torch.rand((n, n), device="cuda"); b = torch.empty_like(a); d = torch.empty_like(a); c = a.clone()
def t(fn, reps=40):
    for _ in range(6): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for x, y in evs:
        x.record(); fn(); y.record()
    torch.cuda.synchronize()
    return statistics.median(x.elapsed_time(y) for x, y in evs) * 1e3
for rnd in range(2):
    for opts in ({}, {"tile_pairs": 32}, {"tile_pairs": 16}, {"tile_pairs": 128}, {"ring": 16}, {"waves": 2}):
        for k, v in opts.items(): dwt.set_option(k, v)
        f = t(lambda: dwt.transform2d_interleaved("cdf97_s", 0, 0, a, b, n*4, 4, n, n, None, None, J))
        i = t(lambda: dwt.transform2d_interleaved("cdf97_s", 1, 0, b, d, n*4, 4, n, n, None, None, J))
        fi = t(lambda: dwt.dwt_cdf97_2f_inplace_s(c, n*4, 4, n, n, n, n, J))
        for k in opts: dwt.set_option(k, {"tile_pairs": 0, "ring": 0, "waves": 4}[k])
        print(f"{str(opts):22s} fwd {f:.1f} us, inv {i:.1f} us, fwd in place {fi:.1f} us", flush=True)
