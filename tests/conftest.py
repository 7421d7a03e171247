import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oraclelib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oraclelib import Reference, have_reference
    if not have_reference():
        pytest.skip("reference library not built here (oracle/_ref absent and no /root/reference)")
    return Reference()


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def golden_cases(wname):
    """[(meta, in, fwd, inv)] for one fixture file; loaded lazily at collection."""
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        man = json.load(f)
    z = np.load(os.path.join(GOLDEN, wname + ".npz"))
    out = []
    for m in man["files"][wname + ".npz"]["cases"]:
        n = m["name"]
        out.append((m, z[n + ".in"], z[n + ".fwd"], z[n + ".inv"]))
    return out


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype.itemsize == 8 else np.uint32)


def interleaved_cases():
    """[(meta, arrays)] of tests/golden/interleaved_s.npz: arrays holds "in" and, per
    wavelet ("cdf97", "cdf53"), "<w>.fwd" / "<w>.inv" (libdwt.h *_inplace_s entries) and,
    for dense frames, "<w>.fdwt2" (dwt-simple.h entries)."""
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        man = json.load(f)
    z = np.load(os.path.join(GOLDEN, "interleaved_s.npz"))
    out = []
    for m in man["files"]["interleaved_s.npz"]["cases"]:
        pre = m["name"] + "."
        out.append((m, {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}))
    return out


def multichannel_cases():
    """[(meta, in, fwd, inv)] of tests/golden/multichannel.npz: (H, pitch_pixels, C) arrays the
    reference transformed one channel of, called the way src/cvdwt.cpp calls the entries."""
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        man = json.load(f)
    z = np.load(os.path.join(GOLDEN, "multichannel.npz"))
    return [(m, z[m["name"] + ".in"], z[m["name"] + ".fwd"], z[m["name"] + ".inv"])
            for m in man["files"]["multichannel.npz"]["cases"]]


def full_range_ints(rng, shape):
    """int32 samples over the WHOLE range: uniform draws mixed with the values where doubled or
    incremented terms wrap (+-2^30 +- k, +-2^31 -+ k, INT_MIN, INT_MAX), the specials also forced
    onto the first / last two rows and columns (the line ends have formulas of their own in the
    reference's int 5/3, src/libdwt.c:10971-10976)."""
    a = rng.integers(-2**31, 2**31, size=shape, dtype=np.int64)
    special = np.array([s * (b + k) for b in (2**30, 2**31 - 8) for k in range(-7, 8) for s in (1, -1)]
                       + [-2**31, 2**31 - 1, -2**31 + 1, 2**31 - 2, 0, 1, -1], dtype=np.int64)
    special = special[(special >= -2**31) & (special < 2**31)]
    pick = rng.random(shape) < 0.3
    a[pick] = rng.choice(special, size=int(pick.sum()))
    h, w = shape
    for r in {0, 1, h - 2, h - 1} & set(range(h)):
        a[r, :] = rng.choice(special, size=w)
    for c in {0, 1, w - 2, w - 1} & set(range(w)):
        a[:, c] = rng.choice(special, size=h)
    return a.astype(np.int32)


def full_range_floats(rng, shape, dtype=np.float32, klass="mixed", nonfinite=False):
    """Float samples over the WHOLE range of the type, the float twin of `full_range_ints`.

    klass: "subnormal"  every sample subnormal (uniform [-1,1) scaled below the smallest normal)
           "tiny"       normals and subnormals mixed around the smallest normal (results underflow
                        gradually in the lifting steps and in the scaling)
           "huge"       magnitudes near the largest finite value, so that lifting steps overflow
                        to +-Inf (and later steps meet Inf - Inf)
           "mixed"      uniform [-1,1) mixed with +-0, subnormals, the smallest / largest normals,
                        values whose doubled tap overflows (> MAX/2) -- the specials also forced onto
                        the first / last two rows and columns (the line ends, where the reference
                        writes 2*c*x: src/libdwt.c:9545-9552, 9873-9907) and onto rows / columns
                        63, 64, 255, 256, 511, 512 (tile seams and the strip rows of the kernels)
    nonfinite: isolated +-Inf and NaNs (quiet and signalling, both signs, payloads) sprinkled in and
        placed on borders and seams as well.
    """
    fi = np.finfo(dtype)
    h, w = shape
    a = (rng.random(shape) * 2 - 1).astype(dtype)
    tiny, big = dtype(fi.tiny), dtype(fi.max)
    sub = dtype(fi.smallest_subnormal)
    if klass == "subnormal":
        return (a * tiny * dtype(0.999)).astype(dtype)
    if klass == "tiny":
        scale = np.where(rng.random(shape) < 0.5, tiny * dtype(4), tiny * dtype(0.25)).astype(dtype)
        return (a * scale).astype(dtype)
    if klass == "huge":
        scale = np.where(rng.random(shape) < 0.5, big, big * dtype(0.25)).astype(dtype)
        return (a * scale).astype(dtype)
    assert klass == "mixed"
    special = np.array([0.0, -0.0, sub, -sub, sub * 3, tiny, -tiny, tiny * dtype(0.75), tiny * dtype(1.5),
                        big, -big, big * dtype(0.5), big * dtype(-0.5), big * dtype(0.6), big * dtype(-0.6),
                        big * dtype(0.3), big * dtype(0.9), 1.0, -1.0, dtype(fi.eps), dtype(1) - dtype(fi.epsneg)], dtype=dtype)
    if nonfinite:
        u = np.uint32 if dtype == np.float32 else np.uint64
        eb = (0xff << 23) if dtype == np.float32 else (0x7ff << 52)
        sb = 1 << (31 if dtype == np.float32 else 63)
        qb = 1 << (22 if dtype == np.float32 else 51)
        nans = np.array([eb | qb, eb | qb | sb, eb | 1, eb | 1 | sb, eb | qb | 0x1234, eb | 0x2a5a5, eb, eb | sb], dtype=u).view(dtype)
        special = np.concatenate([special, nans])
    pick = rng.random(shape) < (0.02 if nonfinite else 0.25)
    a[pick] = rng.choice(special, size=int(pick.sum()))
    rows = ({0, 1, h - 2, h - 1, 7, 8, 13, 14, 63, 64, 127, 128, 255, 256, 511, 512}) & set(range(h))
    cols = ({0, 1, w - 2, w - 1, w - 8, w - 9, 63, 64, 255, 256, 511, 512}) & set(range(w))
    dens = 0.15 if nonfinite else 1.0
    for r in rows:
        m = rng.random(w) < dens
        a[r, m] = rng.choice(special, size=int(m.sum()))
    for c in cols:
        m = rng.random(h) < dens
        a[m, c] = rng.choice(special, size=int(m.sum()))
    return a


def same_floats(got, want):
    """The float parity criterion over the whole range: every sample that is not a NaN on either
    side has the SAME BITS (so +-0, subnormals, +-Inf and every finite value are compared exactly) and
    the NaNs sit at the same positions.  NaN payloads and signs are left out: IEEE 754 does not fix
    which operand's payload an operation with two NaNs returns, x86 SSE returns the first operand's,
    gfx950 its own canonical choice, and the compilers may commute the operands."""
    got, want = np.ascontiguousarray(got), np.ascontiguousarray(want)
    ng, nw = np.isnan(got), np.isnan(want)
    if not np.array_equal(ng, nw):
        return False
    return np.array_equal(bits(got)[~ng], bits(want)[~nw])
