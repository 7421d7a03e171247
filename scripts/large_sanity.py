#!/usr/bin/env python3
"""Large / extreme shapes on the device: forward + inverse round trip and agreement of the
fused path with the exact line-pass path (option generic) on a sub-sampled checksum."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
for (h, w, J, wav, dt) in [(32768, 32768, 5, "cdf97_s", torch.float32), (64, 1 << 20, 3, "cdf97_s", torch.float32),
                           (1 << 20, 64, 3, "cdf97_s", torch.float32), (40000, 3000, -1, "cdf53_i", torch.int32),
                           (3001, 70001, 6, "cdf53_s", torch.float32)]:
    if dt == torch.int32:
        a = torch.randint(-32768, 32768, (h, w), device="cuda", dtype=dt)
    else:
        a = torch.rand((h, w), device="cuda", dtype=dt)
    f = torch.empty_like(a); g = torch.empty_like(a); r = torch.empty_like(a)
    j = dwt._fwd(dwt.WAVELET_ID[wav], a, f, w * 4, 4, w, h, w, h, J, 0, 0, "fwd")
    dwt.set_option("generic", 1)
    j2 = dwt._fwd(dwt.WAVELET_ID[wav], a, g, w * 4, 4, w, h, w, h, J, 0, 0, "fwd")
    dwt.set_option("generic", 0)
    same = torch.equal(f, g)
    dwt._inv(dwt.WAVELET_ID[wav], f, r, w * 4, 4, w, h, w, h, j, 0, 0, "inv")
    torch.cuda.synchronize()
    err = (r.double() - a.double()).abs().max().item()
    print(f"{h}x{w} {wav} J={j}/{j2}: fused == line passes: {same}; round-trip max err {err:.3e}", flush=True)
    del a, f, g, r

# interleaved layout: extreme shapes, the fused sweeps (levels on their lattices, border strips in the launches) vs the
# generic path (the reference's phase order pass by pass), bit for bit both ways; in place (level 0 over the halo
# snapshot) vs out of place
for (h, w, J) in [(16384, 16384, 6), (64, 1 << 20, 3), (1 << 20, 64, 3), (3001, 70001, -1)]:
    a = torch.rand((h, w), device="cuda")
    f = torch.empty_like(a); g = torch.empty_like(a)
    ok = True
    for wav in ("cdf53_s", "cdf97_s"):
        j = dwt.transform2d_interleaved(wav, 0, 0, a, f, w * 4, 4, w, h, None, None, J)
        dwt.set_option("generic", 1)
        dwt.transform2d_interleaved(wav, 0, 0, a, g, w * 4, 4, w, h, None, None, J)
        dwt.set_option("generic", 0)
        ok = ok and torch.equal(f, g)
        g.copy_(a)
        dwt.transform2d_interleaved(wav, 0, 0, g, g, w * 4, 4, w, h, None, None, J)
        ok = ok and torch.equal(f, g)
    # (f, g: the 9/7 coefficients) inverse: out of place, generic, in place
    r0 = torch.empty_like(a)
    dwt.transform2d_interleaved("cdf97_s", 1, 0, f, r0, w * 4, 4, w, h, None, None, j)
    dwt.set_option("generic", 1)
    r1 = torch.empty_like(a)
    dwt.transform2d_interleaved("cdf97_s", 1, 0, f, r1, w * 4, 4, w, h, None, None, j)
    dwt.set_option("generic", 0)
    dwt.transform2d_interleaved("cdf97_s", 1, 0, g, g, w * 4, 4, w, h, None, None, j)
    torch.cuda.synchronize()
    ok_i = torch.equal(r0, r1) and torch.equal(r0, g)
    err = (r0 - a).abs().max().item()
    print(f"{h}x{w} interleaved cdf53_s/cdf97_s J={j}: forward fused == line passes == in place: {ok}; inverse alike: {ok_i}; round-trip max err {err:.3e}", flush=True)
    del a, f, g, r0, r1
# 3-D out of place: the fused one-pass level vs the two-pass path, bit for bit
for (nz, ny, nx, lv) in [(1024, 1024, 1024, 3), (301, 1000, 1111, 2), (2050, 64, 4096, 1)]:
    a = torch.rand((nz, ny, nx), device="cuda")
    f = torch.empty_like(a); g = torch.empty_like(a)
    dwt.set_option("vol_fused", 2)
    dwt.transform3d_op(a, f, nx * 4, nx * ny * 4, nx, ny, nz, lv)
    dwt.set_option("vol_fused", 0)
    dwt.transform3d_op(a, g, nx * 4, nx * ny * 4, nx, ny, nz, lv)
    dwt.set_option("vol_fused", 1)
    same = torch.equal(f, g)
    dwt.transform3d(1, f, nx * 4, nx * ny * 4, nx, ny, nz, lv)
    torch.cuda.synchronize()
    err = (f - a).abs().max().item()
    print(f"{nz}x{ny}x{nx} volume cdf97_s J={lv}: fused == line passes: {same}; round-trip max err {err:.3e}", flush=True)
    del a, f, g
# 3-D in place: the one-pass levels over a halo snapshot (round 3) vs the out-of-place levels (forward) and
# vs two passes through scratch (inverse), bit for bit, at full size and on a ragged volume
for (nz, ny, nx, lv) in [(1024, 1024, 1024, 3), (301, 1000, 1111, 2)]:
    a = torch.rand((nz, ny, nx), device="cuda")
    ref = torch.empty_like(a)
    dwt.transform3d_op(a, ref, nx * 4, nx * ny * 4, nx, ny, nz, lv)
    f = a.clone()
    dwt.transform3d(0, f, nx * 4, nx * ny * 4, nx, ny, nz, lv)
    same_f = torch.equal(f, ref)
    dwt.transform3d(1, f, nx * 4, nx * ny * 4, nx, ny, nz, lv)
    dwt.set_option("vol_inplace_fused", 0)
    dwt.transform3d(1, ref, nx * 4, nx * ny * 4, nx, ny, nz, lv)
    dwt.set_option("vol_inplace_fused", 1)
    torch.cuda.synchronize()
    print(f"{nz}x{ny}x{nx} volume in place J={lv}: one-pass forward == out of place: {same_f}; one-pass inverse == two-pass: {torch.equal(f, ref)}; "
          f"round-trip max err {(f - a).abs().max().item():.3e}", flush=True)
    del a, ref, f
