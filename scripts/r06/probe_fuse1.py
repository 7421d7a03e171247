#!/usr/bin/env python3
"""TIMING PROBE of levels 0 + 1 in one sweep: option probe_fuse1 makes the level-0 kernel run level 1's
arithmetic on its LL rows (neighbours by wavefront shifts, state in registers) and store level 1's four quarter rows instead
of the LL band -- WRONG RESULTS (no halo between tiles, no warm-up above a tile), the cost of the fused kernel without the
halo recomputation a real one needs; probe_fuse1 = 2 adds that geometry (8 more warm-up row pairs per tile, tiles 496 columns
apart with recomputing outer lanes).  Compares, on 8192^2: levels 0 + 1 today (J = 2), level 0 alone (J = 1), the probe."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DWT_HIP_LIB", os.path.join(ROOT, "libdwt_amd", "libdwt_hip_probes.so"))  # make -C libdwt_amd/csrc probes
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n = 8192
def timed(fn, reps):
    for i in range(3): fn(i)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i, (a, b) in enumerate(ev):
        a.record(); fn(i); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) * 1e3 for a, b in ev)
for nb, rot, reps in ((1, 8, 40), (8, 2, 12), (32, 2, 8)):
    src = torch.rand((rot, nb, n, n), device="cuda"); dst = torch.empty_like(src)
    res = {}
    for rnd in range(3):
        for name, J, probe in (("levels 0+1 today", 2, 0), ("level 0 alone", 1, 0), ("probe (0+1 fused, no halo)", 1, 1), ("probe + halo geometry", 1, 2), ("probe + warm-up only", 1, 3), ("probe + 496-column pitch only", 1, 4), ("5 levels today", 5, 0)):
            dwt.set_option("probe_fuse1", probe)
            t = timed(lambda i: dwt.transform2d_batch("cdf97_s", 0, src[i % rot], dst[i % rot], n * n * 4, nb, n * 4, n, n, J), reps)
            if rnd:
                res.setdefault(name, []).append(t)
    dwt.set_option("probe_fuse1", 0)
    m = {k: statistics.median(v) for k, v in res.items()}
    gain = (m["levels 0+1 today"] - m["probe (0+1 fused, no halo)"]) / m["5 levels today"]
    gain2 = (m["levels 0+1 today"] - m["probe + halo geometry"]) / m["5 levels today"]
    print(f"{nb} image(s): " + ", ".join(f"{k} {v:.1f} us" for k, v in m.items()) + f"  -> the five-level call would gain {gain * 100:.1f} % before halo costs, {gain2 * 100:.1f} % with them", flush=True)
    del src, dst
    torch.cuda.empty_cache()
