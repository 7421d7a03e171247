#!/usr/bin/env python3
"""Forward float 9/7, 5 levels, resident batches of ~8 GiB placed by dwt_hip_alloc_batch, for image sizes that do
not fill the 512-column tiles: Gsamples/s and algorithmic GB/s per size (HIP events, median of 10 calls)."""
import os, sys, statistics
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
J = 5
SIZES = [(8192, 8192), (8000, 8000), (2160, 3840), (3001, 4001), (4096, 4096), (5000, 7000), (1080, 1920)]
if len(sys.argv) > 1:
    SIZES = [tuple(int(v) for v in a.split("x"))[::-1] for a in sys.argv[1:]]
for (h, w) in SIZES:
    nb = max(2, (8 << 30) // (h * w * 4))
    src, dst = dwt.alloc_batch("cdf97_s", nb, w, h, J)
    rep = dwt.alloc_batch_report()
    dwt.lib.dwt_hip_probe_pair_us(src, None, nb * h * w * 4)  # finite data
    fn = lambda: dwt.transform2d_batch("cdf97_s", 0, src, dst, h * w * 4, nb, w * 4, w, h, J)
    for _ in range(3): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = statistics.median(a.elapsed_time(b) for a, b in evs)
    extra = ""
    if os.environ.get("AB"):
        name, vals = os.environ["AB"].split("=")
        for v in vals.split(","):
            dwt.set_option(name, int(v))
            for _ in range(3): fn()
            for a, b in evs:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            extra += f" | {name}={v}: {statistics.median(a.elapsed_time(b) for a, b in evs):.3f} ms"
    alg = sum(2 * 4 * -(-w // (1 << j)) * -(-h // (1 << j)) for j in range(J)) * nb
    print(f"{w:5d} x {h:5d} x {nb:4d} images: {ms:8.3f} ms  {nb*h*w/ms/1e6:7.1f} Gsamples/s  {alg/ms/1e6:6.0f} GB/s algorithmic  "
          f"(search: best {rep['whole_call_ms_best_worst'][0]:.3f} worst {rep['whole_call_ms_best_worst'][1]:.3f} ms, {rep['seconds']:.1f} s){extra}", flush=True)
    dwt.lib.dwt_hip_free(src); dwt.lib.dwt_hip_free(dst)
    dwt.dwt_util_finish()
