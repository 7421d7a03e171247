#!/usr/bin/env python3
"""In-process A/B sweep of backend options on the headline workload (one process,
interleaved rounds, medians) -- run on the GPU box:
    python scripts/sweep.py "cpt=8,tile_pairs=32" "cpt=8,tile_pairs=32,nt=1" ...
Prints per variant: median step ms, Gsamples/s, level-0 kernel ms and GB/s."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt

DEFAULTS = dict(generic=0, cpt=0, tile_pairs=0, waves=4, xcd_swizzle=1, ring=0, nt=7, ring_inv=8, fma=0)

def main():
    n = int(os.environ.get("SIZE", 8192)); J = int(os.environ.get("LEVELS", 5)); nb = int(os.environ.get("IMAGES", 4))
    rounds = int(os.environ.get("ROUNDS", 5)); steps = int(os.environ.get("STEPS", 5))
    inverse = int(os.environ.get("INVERSE", 0)); wav = os.environ.get("WAVELET", "cdf97_s")
    variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in sys.argv[1:]] or [{}]
    dwt.dwt_util_init()
    dwt.set_stream(torch.cuda.current_stream().cuda_stream)
    dt = torch.int32 if wav == "cdf53_i" else torch.float32
    if dt == torch.int32:
        src = torch.randint(-32768, 32768, (nb, n, n), device="cuda", dtype=dt)
    else:
        src = torch.rand((nb, n, n), device="cuda", dtype=dt)
    dst = torch.empty_like(src)
    if inverse:
        dwt.transform2d_batch(wav, 0, src, dst, n*n*4, nb, n*4, n, n, J); src, dst = dst, src
    res = {i: {"step": [], "k": []} for i in range(len(variants))}
    for r in range(rounds + 1):
        for i, v in enumerate(variants):
            for k, d in DEFAULTS.items():
                dwt.set_option(k, int(v.get(k, d)))
            dwt.transform2d_batch(wav, inverse, src, dst, n*n*4, nb, n*4, n, n, J)
            torch.cuda.synchronize()
            dwt.prof_enable(True)
            t0 = time.perf_counter()
            for _ in range(steps):
                dwt.transform2d_batch(wav, inverse, src, dst, n*n*4, nb, n*4, n, n, J)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / steps
            ms, cnt = dwt.prof_read(); dwt.prof_enable(False)
            if r > 0:
                res[i]["step"].append(el * 1e3); res[i]["k"].append(ms / max(cnt, 1))
    for i, v in enumerate(variants):
        st = statistics.median(res[i]["step"]); k = statistics.median(res[i]["k"])
        print(f"{sys.argv[1+i] if len(sys.argv)>1 else 'default':50s} step {st:.4f} ms  {nb*n*n/st/1e6:8.1f} Gs/s   L0 {k:.4f} ms {2*4*n*n*nb/k/1e6:8.1f} GB/s  (min step {min(res[i]['step']):.4f})")

if __name__ == "__main__":
    main()
