import os, sys, time, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, J = 8192, 5
a = torch.rand((n, n), device="cuda"); b = torch.empty_like(a)
D = dict(ring=0, tile_pairs=0, wave_horiz=-1, waves=4)
for opts in ["", "ring=16", "tile_pairs=32", "tile_pairs=32,ring=8", "tile_pairs=16", "ring=16,tile_pairs=128", "tile_pairs=128", ""]:
    for k, v in D.items(): dwt.set_option(k, v)
    for kv in [x for x in opts.split(",") if x]:
        k, v = kv.split("="); dwt.set_option(k, int(v))
    res = {}
    for name, fn in (("il", lambda: dwt.transform2d_interleaved("cdf97_s", 0, 0, a, b, n*4, 4, n, n, None, None, J)),
                     ("mallat", lambda: dwt.dwt_cdf97_2f_s2(a, b, n*4, 4, n, n, n, n, J)),
                     ("il_inv", lambda: dwt.transform2d_interleaved("cdf97_s", 1, 0, a, b, n*4, 4, n, n, None, None, J))):
        for _ in range(2): fn()
        torch.cuda.synchronize(); ts = []
        for _ in range(12):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[name] = statistics.median(ts) * 1e6
    print(f"{opts or 'default':28s} il fwd {res['il']:7.1f} us   mallat fwd {res['mallat']:7.1f} us   il inv {res['il_inv']:7.1f} us", flush=True)
