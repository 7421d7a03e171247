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
