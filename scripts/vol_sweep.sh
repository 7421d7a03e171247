python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n=1024
V = torch.rand((n, n, n), device="cuda")
for opts in ["", "cpt=4", "cpt=4,tile_pairs=32", "cpt=4,tile_pairs=128", "cpt=8,tile_pairs=128"]:
    for k,v in dict(cpt=0,tile_pairs=0).items(): dwt.set_option(k,v)
    for kv in [x for x in opts.split(",") if x]:
        k,v=kv.split("="); dwt.set_option(k,int(v))
    for inv in (0,1):
        for _ in range(2): dwt.transform3d(inv, V, n*4, n*n*4, n, n, n, 1)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(5): dwt.transform3d(inv, V, n*4, n*n*4, n, n, n, 1)
        torch.cuda.synchronize(); el=(time.perf_counter()-t0)/5
        print(f"{opts:28s} inv={inv} {el*1e3:.3f} ms {n**3/el/1e9:.1f} Gvox/s")
PY
