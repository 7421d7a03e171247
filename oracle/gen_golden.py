#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference library itself.

TEST INFRASTRUCTURE ONLY.  Run in the build container, where /root/reference exists:

    make -C oracle ref && python oracle/gen_golden.py

Every fixture is DATA: a seeded/synthetic input image, the forward coefficients the
reference (oracle/_ref/libdwt_ref.so, libdwt 2015-02-18-dev built with its own
release flags) produced for it, the level count it returned, and its inverse of
those coefficients.  The whole allocated frame (including margins of sparse frames
and pitch padding) is stored, so tests also pin what the reference leaves untouched.
The reference has no golden vectors of its own: these files are what
pins the oracle and the HIP path on the GPU box, where /root/reference is absent.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oraclelib import Reference  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# (name, size_o (x,y), size_i (x,y) or None, j, decompose_one, zero_padding, pitch_pad_elems, input kind)
CASES_2D = [
    ("8x8_rand", (8, 8), None, -1, 0, 0, 0, "rand"),
    ("16x16_pat", (16, 16), None, -1, 0, 0, 0, "pattern"),
    ("64x64_rand_j3", (64, 64), None, 3, 0, 0, 0, "rand"),
    ("64x64_pat_full", (64, 64), None, -1, 0, 0, 3, "pattern"),
    ("37x53_rand", (37, 53), None, -1, 0, 0, 0, "rand"),
    ("53x37_rand_j2", (53, 37), None, 2, 0, 0, 1, "rand"),
    ("64x5_rand", (64, 5), None, -1, 0, 0, 0, "rand"),
    ("5x64_rand_d1", (5, 64), None, -1, 1, 0, 0, "rand"),
    ("1x64_rand", (1, 64), None, -1, 1, 0, 0, "rand"),
    ("64x1_rand", (64, 1), None, -1, 1, 0, 0, "rand"),
    ("2x2_rand", (2, 2), None, -1, 0, 0, 0, "rand"),
    ("3x3_rand", (3, 3), None, -1, 0, 0, 0, "rand"),
    ("4x4_rand", (4, 4), None, -1, 0, 0, 0, "rand"),
    ("2x7_rand_d1", (2, 7), None, -1, 1, 0, 0, "rand"),
    ("100x100_pat_full", (100, 100), None, -1, 0, 0, 0, "pattern"),
    ("129x65_rand_j4", (129, 65), None, 4, 0, 0, 7, "rand"),
    ("sparse_64x64_50x40", (64, 64), (50, 40), -1, 0, 0, 0, "rand"),
    ("sparse_64x64_50x40_zp", (64, 64), (50, 40), -1, 0, 1, 0, "rand"),
    ("sparse_37x53_30x53_zp_j2", (37, 53), (30, 53), 2, 0, 1, 2, "rand"),
    ("sparse_33x17_20x9", (33, 17), (20, 9), -1, 0, 0, 0, "rand"),
    ("sparse_40x24_1x1_zp", (40, 24), (1, 1), 2, 0, 1, 0, "rand"),
    ("256x192_rand_j5", (256, 192), None, 5, 0, 0, 0, "rand"),
]

WAVELETS = {
    "cdf97_s": ("cdf97_2f_s", "cdf97_2i_s", np.float32),
    "cdf53_i": ("cdf53_2f_i", "cdf53_2i_i", np.int32),
    "cdf53_s": ("cdf53_2f_s", "cdf53_2i_s", np.float32),
    "cdf97_d": ("cdf97_2f_d", "cdf97_2i_d", np.float64),
    "cdf53_d": ("cdf53_2f_d", "cdf53_2i_d", np.float64),
    "cdf97_i": ("cdf97_2f_i", "cdf97_2i_i", np.int32),
}

# the double-precision drivers get a subset of the cases (they run the same drivers'
# geometry code; the fixtures pin the arithmetic and the N==1 / sparse-frame behaviour)
DOUBLE_CASES = {"8x8_rand", "37x53_rand", "64x5_rand", "1x64_rand", "64x1_rand", "2x2_rand", "3x3_rand", "4x4_rand",
                "100x100_pat_full", "sparse_64x64_50x40_zp", "sparse_33x17_20x9", "sparse_40x24_1x1_zp", "129x65_rand_j4"}


# cases of CASES_2D that also pin the interleaved-layout entries
IL_CASES = {"8x8_rand", "16x16_pat", "37x53_rand", "53x37_rand_j2", "64x5_rand", "5x64_rand_d1", "1x64_rand", "2x2_rand",
            "3x3_rand", "4x4_rand", "2x7_rand_d1", "100x100_pat_full", "129x65_rand_j4", "sparse_64x64_50x40",
            "sparse_33x17_20x9", "256x192_rand_j5"}


def make_input(ref, kind, dt, h, w_alloc, w, seed):
    rng = np.random.default_rng(seed)
    if dt == np.float32:
        buf = rng.random((h, w_alloc), dtype=np.float32)
    elif dt == np.float64:
        buf = rng.random((h, w_alloc))
    else:
        buf = rng.integers(-32768, 32768, size=(h, w_alloc), dtype=np.int32)
    if kind == "pattern":
        if dt == np.float32:
            ref.fill_s(buf[:, :w])
        elif dt == np.float64:
            tmp = np.zeros((h, w), np.float32)
            ref.fill_s(tmp)
            buf[:, :w] = tmp
        else:
            ref.fill_i(buf[:, :w])
    return buf


# Multi-channel images as the reference's OpenCV wrapper passes them (src/cvdwt.cpp:98-135):
# ptr = data + elemSize1*channel, stride_x = step, stride_y = elemSize (= channels*elemSize1),
# outer size = the matrix, inner size = the caller's Size, flags -> decompose_one / zero_padding.
# (name, (w, h), channels, channel, size_i or None, j, decompose_one, zero_padding, pitch pad in pixels)
CASES_MC = [
    ("mc3_40x24_c0", (40, 24), 3, 0, None, -1, 0, 0, 0),
    ("mc3_40x24_c2_j2", (40, 24), 3, 2, None, 2, 0, 0, 0),
    ("mc3_37x53_c1", (37, 53), 3, 1, None, -1, 0, 0, 1),
    ("mc2_64x48_c1_sparse_zp", (64, 48), 2, 1, (50, 40), -1, 0, 1, 0),
    ("mc4_33x17_c3_sparse", (33, 17), 4, 3, (20, 9), -1, 0, 0, 0),
    ("mc3_5x64_c1_d1", (5, 64), 3, 1, None, -1, 1, 0, 0),
    ("mc3_96x64_c1_j4", (96, 64), 3, 1, None, 4, 0, 0, 0),
]
# which wavelets run which case (keeps the fixture small): the two north-star wavelets run all
MC_SUBSET = {"cdf53_s": {"mc3_40x24_c0", "mc2_64x48_c1_sparse_zp"}, "cdf97_i": {"mc3_40x24_c2_j2", "mc4_33x17_c3_sparse"},
             "cdf97_d": {"mc3_40x24_c0", "mc4_33x17_c3_sparse"}, "cdf53_d": {"mc3_37x53_c1"}}


def gen_multichannel(ref, manifest):
    arrays, meta = {}, []
    for idx, (name, (w, h), nch, ch, si, j, d1, zp, pad) in enumerate(CASES_MC):
        for wname, (ff, fi, dt) in WAVELETS.items():
            if wname in MC_SUBSET and name not in MC_SUBSET[wname]:
                continue
            rng = np.random.default_rng(6000 + idx)
            shape = (h, w + pad, nch)
            if dt == np.int32:
                buf = rng.integers(-32768, 32768, size=shape, dtype=np.int32)
            else:
                buf = rng.random(shape).astype(dt)
            img = buf[:, :w, :]
            key = f"{name}.{wname}"
            arrays[key + ".in"] = buf.copy()
            jret = ref.call_channel(ff, img, ch, j, size_i=si, decompose_one=d1, zero_padding=zp)
            arrays[key + ".fwd"] = buf.copy()
            ref.call_channel(fi, img, ch, jret, size_i=si, decompose_one=d1, zero_padding=zp)
            arrays[key + ".inv"] = buf.copy()
            meta.append({"name": key, "wavelet": wname, "size_o": (w, h), "size_i": si or (w, h), "channels": nch, "channel": ch,
                         "j_in": j, "j_out": jret, "decompose_one": d1, "zero_padding": zp, "pitch_pixels": w + pad})
    path = os.path.join(OUT, "multichannel.npz")
    np.savez_compressed(path, **arrays)
    manifest["files"]["multichannel.npz"] = {"sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(), "cases": meta}
    print(path, os.path.getsize(path), "bytes", len(meta), "cases")


# 3-D volumes WIDE enough for the fused one-pass kernels (>= 128 samples along x, several tile rows /
# columns, tiles that overhang), through the reference's in-place AND out-of-place forward entries
# (cdf97_3f_ip_sep_horizontal_s, cdf97_3f_op_sep_horizontal_s, src/volume-dwt.c:677, :727) and its
# inverse (:1115); inputs: seeded uniform [0,1) and the reference's own volume_fill_s pattern
# (src/volume.c:41-66).  "full" cases store the arrays; "digest" cases store the sha256 of the
# input and of each output (the volumes would be megabytes each) -- the input is regenerated by the
# test from the same seed / by the product's volume_fill_s and checked against its digest first.
# (name, (nz, ny, nx), input kind, full?)
CASES_VOL = [
    ("vol_10x34x260_rand", (10, 34, 260), "rand", True),
    ("vol_12x40x256_pat", (12, 40, 256), "pattern", True),
    ("vol_24x40x256_rand", (24, 40, 256), "rand", False),
    ("vol_33x35x300_rand", (33, 35, 300), "rand", False),
    ("vol_9x130x513_rand", (9, 130, 513), "rand", False),
    ("vol_37x50x260_pat", (37, 50, 260), "pattern", False),
    ("vol_64x96x512_pat", (64, 96, 512), "pattern", False),
    ("vol_40x70x1030_rand", (40, 70, 1030), "rand", False),
]


def gen_volumes(ref, manifest):
    import ctypes as C

    class Vol(C.Structure):
        _fields_ = [("size_x", C.c_int), ("size_y", C.c_int), ("size_z", C.c_int), ("stride_x", C.c_size_t),
                    ("stride_y", C.c_size_t), ("stride_z", C.c_size_t), ("data", C.c_void_p)]

    def vol(a):
        return Vol(a.shape[2], a.shape[1], a.shape[0], a.strides[2], a.strides[1], a.strides[0], a.ctypes.data)

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

    arrays, meta = {}, []
    for idx, (name, shp, kind, full) in enumerate(CASES_VOL):
        if kind == "rand":
            v = np.random.default_rng(7000 + idx).random(shp, dtype=np.float32)
        else:
            v = np.zeros(shp, np.float32)
            ref.lib.volume_fill_s(C.byref(vol(v)))
        ip = v.copy()
        ref.lib.cdf97_3f_ip_sep_horizontal_s(C.byref(vol(ip)))
        op = np.full(shp, -7.0, np.float32)
        src = v.copy()
        ref.lib.cdf97_3f_op_sep_horizontal_s(C.byref(vol(src)), C.byref(vol(op)))
        assert np.array_equal(src.view(np.uint32), v.view(np.uint32)), "the reference modified its source"
        inv = ip.copy()
        ref.lib.cdf97_3i_ip_sep_horizontal_s(C.byref(vol(inv)))
        m = {"name": name, "shape_zyx": shp, "input": kind, "seed": 7000 + idx if kind == "rand" else None, "full": full,
             "sha256": {"in": sha(v), "fwd": sha(ip), "fwd_op": sha(op), "inv": sha(inv)},
             "op_equals_ip": bool(np.array_equal(ip.view(np.uint32), op.view(np.uint32)))}
        if full:
            arrays[f"{name}.in"] = v
            arrays[f"{name}.fwd"] = ip
            arrays[f"{name}.fwd_op"] = op
            arrays[f"{name}.inv"] = inv
        meta.append(m)
    # the single-direction schedules of the dispatcher (VOL_SEP_HORIZONTAL_X / _Y / _Z = 10, 11, 12,
    # src/volume-dwt.c:788, :852, :918: x copies then lifts, y and z lift the destination in place)
    shp = (9, 11, 14)
    v = np.random.default_rng(7100).random(shp, dtype=np.float32)
    arrays["vol_dirs.in"] = v
    src = v.copy()  # (the struct holds an address only: the array must outlive the call)
    for ap, tag in ((10, "x"), (11, "y"), (12, "z")):
        dst = v.copy() if ap != 10 else np.full(shp, -3.0, np.float32)
        ref.lib.cdf97_3f_op_wrapper_s(C.byref(vol(src)), C.byref(vol(dst)), ap)
        arrays[f"vol_dirs.{tag}"] = dst
    assert np.array_equal(src, v)
    # the pattern itself
    pat = np.zeros((13, 9, 21), np.float32)
    ref.lib.volume_fill_s(C.byref(vol(pat)))
    arrays["vol_fill_13x9x21"] = pat
    path = os.path.join(OUT, "cdf97_3d_wide.npz")
    np.savez_compressed(path, **arrays)
    manifest["files"]["cdf97_3d_wide.npz"] = {"sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(), "cases": meta}
    print(path, os.path.getsize(path), "bytes", len(meta), "cases")


# ---- the whole float range (subnormals, +-0, near-overflow, +-Inf, NaN): tests/conftest.py full_range_floats ----
# (entry pair key, shapes (h, w, j, decompose_one)); inputs are regenerated from the seed by the tests (the
# generator is deterministic), small cases also stored in full; outputs of the larger ones by a sha256 over
# their bits with every NaN replaced by one canonical NaN (payloads are outside the parity criterion)
FR_ENTRIES = {
    "cdf97_s": ("cdf97_2f_s", "cdf97_2i_s", np.float32),
    "cdf53_s": ("cdf53_2f_s", "cdf53_2i_s", np.float32),
    "cdf97_d": ("cdf97_2f_d", "cdf97_2i_d", np.float64),
    "cdf53_d": ("cdf53_2f_d", "cdf53_2i_d", np.float64),
    "cdf97_il": ("cdf97_2f_inplace_s", "cdf97_2i_inplace_s", np.float32),
    "cdf53_il": ("cdf53_2f_inplace_s", "cdf53_2i_inplace_s", np.float32),
}
FR_CLASSES = [("subnormal", 0), ("tiny", 0), ("huge", 0), ("mixed", 0), ("mixed", 1)]
FR_SHAPES = [(37, 53, -1, 0, True), (2, 7, -1, 1, True), (9, 1, -1, 1, True), (64, 80, 2, 0, True),
             (300, 600, 3, 0, False), (515, 1030, -1, 0, False)]
FR_VOLS = [((9, 7, 6), True), ((12, 40, 272), False), ((33, 70, 300), False)]


def canonical_sha(a):
    a = np.ascontiguousarray(a).copy()
    a[np.isnan(a)] = np.nan
    return hashlib.sha256(a.tobytes()).hexdigest()


def gen_float_range(ref, manifest):
    import ctypes as C
    import warnings

    from conftest import full_range_floats

    warnings.simplefilter("ignore")
    arrays, meta = {}, []
    seed = 9000
    for wname, (ff, fi, dt) in FR_ENTRIES.items():
        for klass, nf in FR_CLASSES:
            for (h, w, j, d1, full) in FR_SHAPES:
                if dt == np.float64 and not full:
                    continue
                seed += 1
                src = full_range_floats(np.random.default_rng(seed), (h, w), dt, klass, bool(nf))
                buf = src.copy()
                jret = ref.fwd(ff, buf, j, decompose_one=d1)
                fwd = buf.copy()
                ref.inv(fi, buf, jret, decompose_one=d1)
                name = f"{wname}.{klass}{'_nf' if nf else ''}.{h}x{w}"
                m = {"name": name, "entry": wname, "klass": klass, "nonfinite": nf, "shape": (h, w), "j_in": j, "j_out": jret,
                     "decompose_one": d1, "seed": seed, "full": full,
                     "sha": {"in": canonical_sha(src), "fwd": canonical_sha(fwd), "inv": canonical_sha(buf)}}
                if full:
                    arrays[name + ".in"], arrays[name + ".fwd"], arrays[name + ".inv"] = src, fwd, buf.copy()
                meta.append(m)

    class Vol(C.Structure):
        _fields_ = [("size_x", C.c_int), ("size_y", C.c_int), ("size_z", C.c_int), ("stride_x", C.c_size_t),
                    ("stride_y", C.c_size_t), ("stride_z", C.c_size_t), ("data", C.c_void_p)]

    for klass, nf in FR_CLASSES:
        for shp, full in FR_VOLS:
            seed += 1
            flat = full_range_floats(np.random.default_rng(seed), (shp[0] * shp[1], shp[2]), np.float32, klass, bool(nf))
            v = flat.reshape(shp).copy()
            b = v.copy()
            vs = Vol(shp[2], shp[1], shp[0], b.strides[2], b.strides[1], b.strides[0], b.ctypes.data)
            ref.lib.cdf97_3f_ip_sep_horizontal_s(C.byref(vs))
            f = b.copy()
            ref.lib.cdf97_3i_ip_sep_horizontal_s(C.byref(vs))
            name = f"vol.{klass}{'_nf' if nf else ''}.{shp[0]}x{shp[1]}x{shp[2]}"
            m = {"name": name, "entry": "cdf97_3d", "klass": klass, "nonfinite": nf, "shape": shp, "seed": seed, "full": full,
                 "sha": {"in": canonical_sha(v), "fwd": canonical_sha(f), "inv": canonical_sha(b)}}
            if full:
                arrays[name + ".in"], arrays[name + ".fwd"], arrays[name + ".inv"] = v, f, b.copy()
            meta.append(m)
    path = os.path.join(OUT, "float_range.npz")
    np.savez_compressed(path, **arrays)
    manifest["files"]["float_range.npz"] = {"sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(), "cases": meta}
    print(path, os.path.getsize(path), "bytes", len(meta), "cases")


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = Reference()
    if sys.argv[1:] == ["volumes"]:
        with open(os.path.join(OUT, "manifest.json")) as f:
            manifest = json.load(f)
        gen_volumes(ref, manifest)
        with open(os.path.join(OUT, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1)
        return
    if sys.argv[1:] == ["float_range"]:
        with open(os.path.join(OUT, "manifest.json")) as f:
            manifest = json.load(f)
        gen_float_range(ref, manifest)
        with open(os.path.join(OUT, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1)
        return
    if sys.argv[1:] == ["multichannel"]:
        # add / refresh this one file, leaving the others (and their hashes) as they are
        with open(os.path.join(OUT, "manifest.json")) as f:
            manifest = json.load(f)
        gen_multichannel(ref, manifest)
        with open(os.path.join(OUT, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1)
        return
    manifest = {"generator": "oracle/gen_golden.py", "reference": "libdwt 2015-02-18-dev (oracle/_ref/libdwt_ref.so)",
                "files": {}}
    for wname, (ff, fi, dt) in WAVELETS.items():
        arrays = {}
        meta = []
        for idx, (name, so, si, j, d1, zp, pad, kind) in enumerate(CASES_2D):
            if dt == np.float64 and name not in DOUBLE_CASES:
                continue
            w, h = so
            buf = make_input(ref, kind, dt, h, w + pad, w, seed=1000 + idx)
            src = buf.copy()
            jret = ref.fwd(ff, buf[:, :w], j, size_o=so, size_i=si, decompose_one=d1, zero_padding=zp)
            fwd = buf.copy()
            ref.inv(fi, buf[:, :w], jret, size_o=so, size_i=si, decompose_one=d1, zero_padding=zp)
            inv = buf.copy()
            arrays[f"{name}.in"] = src
            arrays[f"{name}.fwd"] = fwd
            arrays[f"{name}.inv"] = inv
            meta.append({"name": name, "size_o": so, "size_i": si or so, "j_in": j, "j_out": jret,
                         "decompose_one": d1, "zero_padding": zp, "pitch_elems": w + pad, "input": kind})
        # out-of-place entries (_s2) for the float 9/7 pair
        if wname == "cdf97_s":
            for idx, (name, so, j) in enumerate([("s2_64x48_rand", (64, 48), -1), ("s2_37x53_rand_j2", (37, 53), 2),
                                                 ("s2_1x9_rand", (1, 9), -1)]):
                w, h = so
                rng = np.random.default_rng(2000 + idx)
                src = rng.random((h, w), dtype=np.float32)
                dst = np.full((h, w), 7.0, np.float32)
                jret = ref.call2("cdf97_2f_s2", src, dst, j)
                rec = np.full((h, w), 3.0, np.float32)
                ref.call2("cdf97_2i_s2", dst, rec, jret)
                arrays[f"{name}.in"] = src
                arrays[f"{name}.fwd"] = dst
                arrays[f"{name}.inv"] = rec
                meta.append({"name": name, "size_o": so, "size_i": so, "j_in": j, "j_out": jret, "s2": True,
                             "dst_fill": 7.0, "rec_fill": 3.0})
        path = os.path.join(OUT, f"{wname}.npz")
        np.savez_compressed(path, **arrays)
        sha = hashlib.sha256(open(path, "rb").read()).hexdigest()
        manifest["files"][f"{wname}.npz"] = {"sha256": sha, "cases": meta}
        print(path, os.path.getsize(path), "bytes", len(meta), "cases")

    # interleaved (in-place lifting) layout: dwt-simple.h fdwt2_* and libdwt.h *_inplace_s
    ref.lib.dwt_util_set_num_workers(1)
    arrays, meta = {}, []
    il_cases = [c for c in CASES_2D if c[0] in IL_CASES] + [
        ("5x5_rand", (5, 5), None, -1, 0, 0, 0, "rand"), ("6x6_rand", (6, 6), None, -1, 0, 0, 0, "rand"),
        ("17x1_rand_d1", (17, 1), None, -1, 1, 0, 0, "rand"), ("12x4_rand_d1", (12, 4), None, -1, 1, 0, 2, "rand"),
    ]
    for idx, (name, so, si, j, d1, zp, pad, kind) in enumerate(il_cases):
        w, h = so
        src = make_input(ref, kind, np.float32, h, w + pad, w, seed=4000 + idx)
        arrays[f"{name}.in"] = src
        m = {"name": name, "size_o": so, "size_i": si or so, "j_in": j, "decompose_one": d1, "pitch_elems": w + pad, "input": kind}
        for wv in ("cdf97", "cdf53"):
            buf = src.copy()
            m[f"{wv}.j_out"] = ref.fwd(f"{wv}_2f_inplace_s", buf[:, :w], j, size_o=so, size_i=si, decompose_one=d1)
            arrays[f"{name}.{wv}.fwd"] = buf.copy()
            ref.inv(f"{wv}_2i_inplace_s", buf[:, :w], m[f"{wv}.j_out"], size_o=so, size_i=si, decompose_one=d1)
            arrays[f"{name}.{wv}.inv"] = buf.copy()
            if si is None:
                outs = []
                for sched in ("horizontal", "vertical", "diagonal"):
                    buf = src.copy()
                    jn = ref.fdwt2(wv, buf[:, :w], j, d1, sched)
                    outs.append(buf)
                    assert jn == m[f"{wv}.j_out"]
                assert all(np.array_equal(outs[0].view(np.uint32), o.view(np.uint32)) for o in outs[1:])
                arrays[f"{name}.{wv}.fdwt2"] = outs[0]
        # fixed-point int 9/7 of the same layout family (src/libdwt.c:17424, 17308), own int input
        isrc = make_input(ref, kind, np.int32, h, w + pad, w, seed=5000 + idx)
        arrays[f"{name}.cdf97i.in"] = isrc
        buf = isrc.copy()
        m["cdf97i.j_out"] = ref.fwd("cdf97_2f_inplace_i", buf[:, :w], j, size_o=so, size_i=si, decompose_one=d1)
        arrays[f"{name}.cdf97i.fwd"] = buf.copy()
        ref.inv("cdf97_2i_inplace_i", buf[:, :w], m["cdf97i.j_out"], size_o=so, size_i=si, decompose_one=d1)
        arrays[f"{name}.cdf97i.inv"] = buf.copy()
        meta.append(m)
    path = os.path.join(OUT, "interleaved_s.npz")
    np.savez_compressed(path, **arrays)
    manifest["files"]["interleaved_s.npz"] = {"sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(), "cases": meta}
    print(path, os.path.getsize(path), "bytes", len(meta), "cases")

    # 3-D single-level float 9/7 (volume-dwt.c sep_horizontal), interleaved layout
    import ctypes as C

    class Vol(C.Structure):
        _fields_ = [("size_x", C.c_int), ("size_y", C.c_int), ("size_z", C.c_int), ("stride_x", C.c_size_t),
                    ("stride_y", C.c_size_t), ("stride_z", C.c_size_t), ("data", C.c_void_p)]

    arrays, meta = {}, []
    for idx, shp in enumerate([(8, 8, 8), (5, 5, 5), (9, 7, 6), (17, 33, 20)]):
        rng = np.random.default_rng(3000 + idx)
        v = rng.random(shp, dtype=np.float32)
        b = v.copy()
        vs = Vol(shp[2], shp[1], shp[0], b.strides[2], b.strides[1], b.strides[0], b.ctypes.data)
        ref.lib.cdf97_3f_ip_sep_horizontal_s(C.byref(vs))
        f = b.copy()
        ref.lib.cdf97_3i_ip_sep_horizontal_s(C.byref(vs))
        name = "vol_%dx%dx%d" % shp
        arrays[f"{name}.in"] = v
        arrays[f"{name}.fwd"] = f
        arrays[f"{name}.inv"] = b.copy()
        meta.append({"name": name, "shape_zyx": shp})
    path = os.path.join(OUT, "cdf97_3d_s.npz")
    np.savez_compressed(path, **arrays)
    manifest["files"]["cdf97_3d_s.npz"] = {"sha256": hashlib.sha256(open(path, "rb").read()).hexdigest(), "cases": meta}
    print(path, os.path.getsize(path), "bytes")

    gen_multichannel(ref, manifest)
    gen_volumes(ref, manifest)
    gen_float_range(ref, manifest)

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
