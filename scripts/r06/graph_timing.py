#!/usr/bin/env python3
"""One 8192^2 image, J = 5: the `_s2` forward / inverse calls eager against the same calls captured into a HIP graph and
replayed (HIP-event medians, inputs rotating over 8 images eager; the graph replays ONE image's call)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init()
side = torch.cuda.Stream()
dwt.set_stream(side.cuda_stream)
n, J, nb = 8192, 5, 8
with torch.cuda.stream(side):
    img = torch.rand((nb, n, n), device="cuda"); out = torch.empty_like(img)
    def timed(fn, reps=80):
        for i in range(5): fn(i)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for i, (a, b) in enumerate(ev):
            a.record(side); fn(i); b.record(side)
        side.synchronize()
        t = [a.elapsed_time(b) * 1e3 for a, b in ev]
        return statistics.median(t), min(t)
    for name, call in (("2f_s2", lambda i: dwt.dwt_cdf97_2f_s2(img[i % nb], out[i % nb], n * 4, 4, n, n, n, n, J)),
                       ("2i_s2", lambda i: dwt.dwt_cdf97_2i_s2(img[i % nb], out[i % nb], n * 4, 4, n, n, n, n, J))):
        call(0); side.synchronize()
        graphs = []
        for k in range(nb):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                call(k)
            graphs.append(g)
        e = timed(call)
        r = timed(lambda i: graphs[i % nb].replay())
        print(f"{name}: eager median {e[0]:.1f} us min {e[1]:.1f} | graph replay median {r[0]:.1f} us min {r[1]:.1f}", flush=True)
dwt.set_stream(0)
