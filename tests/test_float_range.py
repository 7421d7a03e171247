"""CPU-only: float parity over the WHOLE range of the type -- subnormals, +-0, magnitudes next to
overflow, +-Inf, quiet and signalling NaNs -- the float twin of the whole-int32-range tests.

Criterion (`conftest.same_floats`): every sample that is not a NaN has the same bits on both sides
(+-0, subnormals, +-Inf included) and the NaNs sit at the same positions; on finite results that
is plain bit equality.  The reference's own comparison calls any NaN / Inf "differs"
(src/libdwt.c:1604-1616): it has no opinion on payloads.

The one thing these tests found: the reference writes the two equal taps of a line end as 2*c*x
(src/libdwt.c:9545-9552, 9873-9907; src/dwt-simple.c:596-603), the restatement used c*(x+x).  Same
bits unless x+x overflows.  The oracle now follows the reference; `Oracle.reflected_ends()` gives the
other form for the GPU tests (the kernels reflect their load addresses)."""
import warnings

import numpy as np
import pytest

from conftest import GOLDEN, bits, full_range_floats, same_floats
from oraclelib import ENTRIES

warnings.filterwarnings("ignore", category=RuntimeWarning)

ENTRY = {
    "cdf97_s": ("cdf97_2f_s", "cdf97_2i_s"),
    "cdf53_s": ("cdf53_2f_s", "cdf53_2i_s"),
    "cdf97_d": ("cdf97_2f_d", "cdf97_2i_d"),
    "cdf53_d": ("cdf53_2f_d", "cdf53_2i_d"),
    "cdf97_il": ("cdf97_2f_inplace_s", "cdf97_2i_inplace_s"),
    "cdf53_il": ("cdf53_2f_inplace_s", "cdf53_2i_inplace_s"),
}
CLASSES = [("subnormal", 0), ("tiny", 0), ("huge", 0), ("mixed", 0), ("mixed", 1)]


def canonical_sha(a):
    import hashlib
    a = np.ascontiguousarray(a).copy()
    a[np.isnan(a)] = np.nan
    return hashlib.sha256(a.tobytes()).hexdigest()


def float_range_cases():
    import json
    import os
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)["files"]["float_range.npz"]["cases"]


def float_range_input(m):
    """The input of a fixture case, regenerated from its seed (checked against the stored hash)."""
    if m["entry"] == "cdf97_3d":
        z, y, x = m["shape"]
        a = full_range_floats(np.random.default_rng(m["seed"]), (z * y, x), np.float32, m["klass"], bool(m["nonfinite"]))
        a = a.reshape(z, y, x).copy()
    else:
        dt = ENTRIES[ENTRY[m["entry"]][0]][1]
        a = full_range_floats(np.random.default_rng(m["seed"]), tuple(m["shape"]), dt, m["klass"], bool(m["nonfinite"]))
    assert canonical_sha(a) == m["sha"]["in"], "the input generator changed: regenerate tests/golden/float_range.npz"
    return a


@pytest.fixture(scope="module")
def stored():
    import os
    return np.load(os.path.join(GOLDEN, "float_range.npz"))


@pytest.mark.parametrize("m", float_range_cases(), ids=lambda m: m["name"])
def test_oracle_matches_the_float_range_fixtures(oracle, stored, m):
    """Outputs of the reference itself (oracle/gen_golden.py float_range) for inputs over the whole
    float range: small cases stored in full, larger ones as a hash over NaN-canonical bits."""
    a = float_range_input(m)
    if m["entry"] == "cdf97_3d":
        fwd = oracle.vol("cdf97_3f_s", a.copy())
        inv = oracle.vol("cdf97_3i_s", fwd.copy())
    else:
        ff, fi = ENTRY[m["entry"]]
        fwd = a.copy()
        assert oracle.fwd(ff, fwd, m["j_in"], decompose_one=m["decompose_one"]) == m["j_out"]
        inv = fwd.copy()
        oracle.inv(fi, inv, m["j_out"], decompose_one=m["decompose_one"])
    if m["full"]:
        assert np.array_equal(bits(a), bits(stored[m["name"] + ".in"]))
        assert same_floats(fwd, stored[m["name"] + ".fwd"])
        assert same_floats(inv, stored[m["name"] + ".inv"])
    assert canonical_sha(fwd) == m["sha"]["fwd"]
    assert canonical_sha(inv) == m["sha"]["inv"]


@pytest.mark.parametrize("klass,nf", CLASSES, ids=lambda v: str(v))
@pytest.mark.parametrize("wname", list(ENTRY))
def test_oracle_equals_reference_over_the_whole_float_range(oracle, reference, wname, klass, nf):
    """Seeded sweep against the compiled reference: shapes with odd / even / short / single lines,
    one level and the full depth, forward, and the inverse of the reference's own forward."""
    ff, fi = ENTRY[wname]
    dt = ENTRIES[ff][1]
    rng = np.random.default_rng(77)
    for (h, w) in [(2, 2), (3, 5), (5, 4), (8, 8), (37, 53), (64, 65), (130, 97), (1, 9), (9, 1), (4, 2)]:
        for j, d1 in ((1, 0), (-1, 0), (-1, 1)):
            a = full_range_floats(rng, (h, w), dt, klass, bool(nf))
            b = a.copy()
            assert oracle.fwd(ff, a, j, decompose_one=d1) == reference.fwd(ff, b, j, decompose_one=d1)
            assert same_floats(a, b), (ff, klass, nf, h, w, j, d1)
            if not np.isnan(b).any():
                assert np.array_equal(bits(a), bits(b))
            a = b.copy()
            oracle.inv(fi, a, j, decompose_one=d1)
            reference.inv(fi, b, j, decompose_one=d1)
            assert same_floats(a, b), (fi, klass, nf, h, w, j, d1)


@pytest.mark.parametrize("klass,nf", CLASSES, ids=lambda v: str(v))
def test_oracle_s2_fdwt2_and_sparse_frames_over_the_whole_float_range(oracle, reference, klass, nf):
    """The out-of-place entries, the dwt-simple.h entries and sparse frames with zero padding."""
    rng = np.random.default_rng(78)
    for (h, w) in [(37, 53), (64, 40), (6, 5)]:
        src = full_range_floats(rng, (h, w), np.float32, klass, bool(nf))
        do, dr = np.full_like(src, 7.0), np.full_like(src, 7.0)
        jo = oracle.call2("cdf97_2f_s2", src.copy(), do, -1)
        assert jo == reference.call2("cdf97_2f_s2", src.copy(), dr, -1) and same_floats(do, dr)
        ro, rr = np.full_like(src, 3.0), np.full_like(src, 3.0)
        oracle.call2("cdf97_2i_s2", dr, ro, jo)
        reference.call2("cdf97_2i_s2", dr, rr, jo)
        assert same_floats(ro, rr)
        for wv in ("cdf97", "cdf53"):
            a, b = src.copy(), src.copy()
            assert oracle.fdwt2(wv, a, -1) == reference.fdwt2(wv, b, -1) and same_floats(a, b), (wv, h, w)
    for ff in ("cdf97_2f_s", "cdf53_2f_s"):
        a = full_range_floats(rng, (53, 64), np.float32, klass, bool(nf))
        b = a.copy()
        kw = dict(size_o=(64, 53), size_i=(50, 40), zero_padding=1)
        assert oracle.fwd(ff, a, -1, **kw) == reference.fwd(ff, b, -1, **kw) and same_floats(a, b)


@pytest.mark.parametrize("klass,nf", CLASSES, ids=lambda v: str(v))
def test_oracle_3d_equals_reference_over_the_whole_float_range(oracle, reference, klass, nf):
    import ctypes as C

    class Vol(C.Structure):
        _fields_ = [("size_x", C.c_int), ("size_y", C.c_int), ("size_z", C.c_int), ("stride_x", C.c_size_t),
                    ("stride_y", C.c_size_t), ("stride_z", C.c_size_t), ("data", C.c_void_p)]

    rng = np.random.default_rng(79)
    for shp in [(5, 5, 5), (9, 7, 6), (6, 11, 40), (17, 8, 33)]:
        v = full_range_floats(rng, (shp[0] * shp[1], shp[2]), np.float32, klass, bool(nf)).reshape(shp).copy()
        b = v.copy()
        vs = Vol(shp[2], shp[1], shp[0], b.strides[2], b.strides[1], b.strides[0], b.ctypes.data)
        reference.lib.cdf97_3f_ip_sep_horizontal_s(C.byref(vs))
        a = oracle.vol("cdf97_3f_s", v.copy())
        assert same_floats(a, b), (shp, klass, nf)
        reference.lib.cdf97_3i_ip_sep_horizontal_s(C.byref(vs))
        assert same_floats(oracle.vol("cdf97_3i_s", a), b), (shp, klass, nf)


def test_the_two_end_forms_differ_only_where_a_doubled_tap_overflows(oracle, reference):
    """(2c)*x (the reference, the oracle's default) against c*(x+x) (reflection; the HIP kernels):
    a known line where they differ, and bit equality on every class whose magnitudes stay below
    half the largest float, non-finite inputs included (2c*Inf == c*(Inf+Inf))."""
    big = np.float32(np.finfo(np.float32).max)
    line = np.zeros((1, 8), np.float32)
    line[0, 1] = big * np.float32(0.75)  # the tap of sample 0 in the update steps
    want = line.copy()
    reference.fwd("cdf97_2f_s", want, 1, decompose_one=1)
    got = line.copy()
    oracle.fwd("cdf97_2f_s", got, 1, decompose_one=1)
    assert np.array_equal(bits(got), bits(want)) and np.isfinite(want[0, 0])
    with oracle.reflected_ends():
        refl = line.copy()
        oracle.fwd("cdf97_2f_s", refl, 1, decompose_one=1)
    assert np.isinf(refl[0, 0]) or np.isnan(refl[0, 0])  # beta * (x + x): x + x overflowed
    rng = np.random.default_rng(80)
    for ff, fi in (ENTRY["cdf97_s"], ENTRY["cdf97_il"], ENTRY["cdf53_il"]):
        for klass, nf, scale in (("subnormal", 0, 1), ("tiny", 0, 1), ("mixed", 1, 2.0 ** -12), ("huge", 0, 2.0 ** -12)):
            for (h, w) in [(37, 53), (64, 65), (2, 9)]:
                a = full_range_floats(rng, (h, w), np.float32, klass, bool(nf)) * np.float32(scale)
                b = a.copy()
                j = oracle.fwd(ff, a, 2, decompose_one=1)
                with oracle.reflected_ends():
                    oracle.fwd(ff, b, 2, decompose_one=1)
                assert same_floats(a, b), (ff, klass, h, w)
                oracle.inv(fi, a, j, decompose_one=1)
                with oracle.reflected_ends():
                    oracle.inv(fi, b, j, decompose_one=1)
                assert same_floats(a, b), (fi, klass, h, w)
