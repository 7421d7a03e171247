#!/usr/bin/env python3
"""A device-pointer call is a pure sequence of launches on the caller's stream: captured into a HIP graph (after one warm-up
call that allocated the scratch) and replayed, it gives the bits of the eager call -- forward and inverse batches, the `_s2`
entry on one image, the interleaved entry.  Prints 'graph replay: N cases, 0 mismatches'.  (Also a GPU test.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt

dwt.dwt_util_init()
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev)
bad = cases = 0


def check(name, run, outputs, poison):
    """eager on the side stream -> keep; poison the outputs; capture + replay -> compare"""
    global bad, cases
    dwt.set_stream(side.cuda_stream)
    with torch.cuda.stream(side):
        run()          # warm-up: allocates the scratch
        side.synchronize()
        want = [o.clone() for o in outputs]
        for o in outputs:
            poison(o)
        launches = dwt.get_option("stat_launches")
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            run()
        recorded = dwt.get_option("stat_launches") - launches
        for o in outputs:  # the capture itself ran nothing
            poison(o)
        g.replay()
        side.synchronize()
    ok = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(outputs, want))
    cases += 1
    bad += 0 if ok else 1
    print(f"{name}: {recorded} launches recorded, replay {'==' if ok else '!='} eager", flush=True)


nb, n, J = 4, 2048, 4
src = torch.rand((nb, n, n), device=dev)
dst = torch.empty_like(src)
rec = torch.empty_like(src)
check("forward batch", lambda: dwt.transform2d_batch("cdf97_s", 0, src, dst, n * n * 4, nb, n * 4, n, n, J), [dst], lambda o: o.fill_(7))
check("inverse batch", lambda: dwt.transform2d_batch("cdf97_s", 1, dst, rec, n * n * 4, nb, n * 4, n, n, J), [rec], lambda o: o.fill_(7))
isrc = torch.randint(-32768, 32768, (nb, n, n), device=dev, dtype=torch.int32)
idst = torch.empty_like(isrc)
check("int 5/3 forward batch", lambda: dwt.transform2d_batch("cdf53_i", 0, isrc, idst, n * n * 4, nb, n * 4, n, n, J), [idst], lambda o: o.fill_(7))
one = torch.empty((n, n), device=dev)
check("dwt_cdf97_2f_s2, one image", lambda: dwt.dwt_cdf97_2f_s2(src[1], one, n * 4, 4, n, n, n, n, J), [one], lambda o: o.fill_(7))
il = torch.empty((n, n), device=dev)
check("interleaved forward", lambda: dwt.transform2d_interleaved("cdf97_s", 0, 0, src[2], il, n * 4, 4, n, n, n, n, J), [il], lambda o: o.fill_(7))
dwt.set_stream(0)
print(f"graph replay: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
