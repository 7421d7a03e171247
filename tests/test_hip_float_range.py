"""GPU parity over the WHOLE float range (subnormals, +-0, near-overflow, +-Inf, quiet / signalling
NaNs on borders, tile seams and strip rows): the HIP path against the oracle -- which the CPU suite
pins to the reference on the same inputs (tests/test_float_range.py, tests/golden/float_range.npz).

Criterion `same_floats`: same bits wherever neither side is a NaN, NaNs at the same positions.

Line ends.  The reference adds (2c)*x where both taps of a sample are the one sample x
(src/libdwt.c:9545-9552, 9873-9907); reflection alone -- which the kernels apply to their load addresses -- would give
c*(x+x): `v_add_f32 t, x, x` overflows to Inf for |x| > FLT_MAX/2 where (2c)*x with |2c| < 1 stays finite.  Every 2-D
kernel of the library applies the reference's own form (dwt_lift.h: SelEnds in the tile sweeps, fwd_end / inv_end
elsewhere): the 2-D entries are compared with the FAITHFUL oracle and the reference's own outputs (the fixtures) on
every class.  The 3-D level kernels keep the reflected form, the one listed difference: the 3-D entries are
compared with the oracle's reflected-ends form on the two classes that reach that range ("huge", "mixed") and with the
fixtures on the others.  `make plain` (libdwt_hip_plain.so, DWT_FLOAT_END_FORMS=0: an A/B timing build, reflected ends
in every float kernel) can be put under this file with DWT_HIP_LIB; it is then held to the reflected form throughout."""
import warnings

import numpy as np
import pytest

from conftest import bits, full_range_floats, same_floats
from test_float_range import ENTRY, float_range_cases, float_range_input

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore", category=RuntimeWarning)

OVERFLOWING = {"huge", "mixed"}
# Which entries evaluate a line end as c*(x+x) (reflected load addresses) depends on the BUILD (dwt_lift.h,
# DWT_FLOAT_END_FORMS): the library only in its 3-D level kernels; the A/B build `make plain` in every float / double entry.
import os

PLAIN_BUILD = os.path.basename(os.environ.get("DWT_HIP_LIB", "")).startswith("libdwt_hip_plain")
ENDS_REFLECTED = {"cdf97_s", "cdf53_s", "cdf97_d", "cdf53_d", "cdf97_il", "cdf53_il", "cdf97_3d"} if PLAIN_BUILD else {"cdf97_3d"}
WID = {"cdf97_s": "cdf97_s", "cdf53_s": "cdf53_s", "cdf97_d": "cdf97_d", "cdf53_d": "cdf53_d"}


@pytest.fixture(scope="module")
def dwt():
    import libdwt_amd as d

    d.dwt_util_init()
    yield d
    for k, v in (("fma", 0), ("vol_fused", 1), ("vol_tile_pairs", 0), ("il_exact_borders", 1)):
        d.set_option(k, v)
    d.dwt_util_set_accel(0)
    d.dwt_util_finish()


@pytest.fixture(scope="module")
def stored():
    import os

    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "float_range.npz"))


def expected(oracle, m, a):
    """(forward, inverse of that forward) the kernels must produce for input `a` of fixture case `m`."""
    refl = m["entry"] in ENDS_REFLECTED and m["klass"] in OVERFLOWING
    ctx = oracle.reflected_ends() if refl else warnings.catch_warnings()
    with ctx:
        if m["entry"] == "cdf97_3d":
            fwd = oracle.vol("cdf97_3f_s", a.copy())
            return fwd, oracle.vol("cdf97_3i_s", fwd.copy())
        ff, fi = ENTRY[m["entry"]]
        fwd = a.copy()
        j = oracle.fwd(ff, fwd, m["j_in"], decompose_one=m["decompose_one"])
        inv = fwd.copy()
        oracle.inv(fi, inv, j, decompose_one=m["decompose_one"])
        return fwd, inv


def gpu_2d(dwt, m, a, fwd_in=None):
    """Forward of `a` and inverse of `fwd_in` through the libdwt.h host-pointer entries."""
    h, w = a.shape
    es = a.dtype.itemsize
    e = m["entry"]
    if e.endswith("_il"):
        f = getattr(dwt, "dwt_%s_2f_inplace_s" % e[:5])
        i = getattr(dwt, "dwt_%s_2i_inplace_s" % e[:5])
    else:
        f, i = dwt.FORWARD[e], dwt.INVERSE[e]
    got = a.copy()
    j = f(got, got.strides[0], es, w, h, w, h, m["j_in"], m["decompose_one"])
    assert j == m["j_out"]
    back = (got if fwd_in is None else fwd_in).copy()
    i(back, back.strides[0], es, w, h, w, h, j, m["decompose_one"])
    return got, back


CASES_2D = [m for m in float_range_cases() if m["entry"] != "cdf97_3d"]
CASES_3D = [m for m in float_range_cases() if m["entry"] == "cdf97_3d"]


@pytest.mark.parametrize("accel", [0, 1], ids=["fused", "line-passes"])
@pytest.mark.parametrize("m", CASES_2D, ids=lambda m: m["name"])
def test_2d_entries_over_the_whole_float_range(dwt, oracle, stored, m, accel):
    """Every float 2-D entry (Mallat 9/7 and 5/3, double, interleaved 9/7 and 5/3), fused sweeps and
    (accel 1) the line passes, forward and inverse."""
    a = float_range_input(m)
    want_f, want_i = expected(oracle, m, a)
    dwt.dwt_util_set_accel(accel)
    try:
        got_f, got_i = gpu_2d(dwt, m, a, fwd_in=want_f)
    finally:
        dwt.dwt_util_set_accel(0)
    assert same_floats(got_f, want_f), "forward"
    assert same_floats(got_i, want_i), "inverse"
    if not (m["entry"] in ENDS_REFLECTED and m["klass"] in OVERFLOWING) and m["full"]:
        # ... and these are the reference's own outputs
        assert same_floats(got_f, stored[m["name"] + ".fwd"])
        if not np.isnan(want_f).any():
            assert np.array_equal(bits(got_f), bits(stored[m["name"] + ".fwd"]))


@pytest.mark.parametrize("m", [m for m in CASES_2D if m["entry"] == "cdf97_s" and m["shape"][0] >= 37], ids=lambda m: m["name"])
def test_s2_and_device_entries_over_the_whole_float_range(dwt, oracle, m):
    """The out-of-place entries on device-resident images (no staging of the detail bands) and the
    in-place device entry (staged, copy riding along)."""
    a = float_range_input(m)
    h, w = a.shape
    want_f, want_i = expected(oracle, m, a)
    src, dst = dwt.DeviceImage(h, w).upload(a), dwt.DeviceImage(h, w).upload(np.zeros_like(a))
    j = dwt.dwt_cdf97_2f_s2(src.ptr, dst.ptr, dst.stride_x, 4, w, h, w, h, m["j_in"], m["decompose_one"])
    assert j == m["j_out"] and same_floats(dst.download(np.float32), want_f)
    src.upload(want_f)
    dwt.dwt_cdf97_2i_s2(src.ptr, dst.ptr, dst.stride_x, 4, w, h, w, h, j, m["decompose_one"])
    assert same_floats(dst.download(np.float32), want_i)
    src.upload(a)
    dwt.dwt_cdf97_2f_s(src.ptr, src.stride_x, 4, w, h, w, h, m["j_in"], m["decompose_one"])
    assert same_floats(src.download(np.float32), want_f)
    src.upload(want_f)
    dwt.dwt_cdf97_2i_s(src.ptr, src.stride_x, 4, w, h, w, h, j, m["decompose_one"])
    assert same_floats(src.download(np.float32), want_i)
    src.free()
    dst.free()


@pytest.mark.parametrize("m", [m for m in CASES_2D if m["entry"] == "cdf97_s"], ids=lambda m: m["name"])
def test_fma_option_over_the_whole_float_range(dwt, oracle, m):
    """Option `fma` contracts each lifting step: not the reference's rounding.  On finite results
    <= 1e-5 relative (to the largest coefficient of the band of values compared); the non-finite
    samples have the same support unless a contracted step does not overflow where mul-then-add did."""
    a = float_range_input(m)
    want_f, _ = expected(oracle, m, a)
    dwt.set_option("fma", 1)
    try:
        got_f, _ = gpu_2d(dwt, m, a, fwd_in=want_f)
    finally:
        dwt.set_option("fma", 0)
    if m["klass"] in OVERFLOWING:
        # a fused step rounds once: c + k*t can stay finite where the rounded product k*t overflowed
        both = np.isfinite(got_f) & np.isfinite(want_f)
        assert both.sum() >= 0.5 * np.isfinite(want_f).sum()
    else:
        assert np.array_equal(np.isnan(got_f), np.isnan(want_f))
        both = np.isfinite(want_f)
        assert np.array_equal(np.isfinite(got_f), both)
    if both.any():
        g, w_ = got_f[both].astype(np.float64), want_f[both].astype(np.float64)
        scale = max(np.abs(w_).max(), float(np.finfo(np.float32).tiny))
        assert np.abs(g - w_).max() <= 1e-5 * scale


@pytest.mark.parametrize("mode", ["op", "ip", "ip-fused"])
@pytest.mark.parametrize("m", CASES_3D, ids=lambda m: m["name"])
def test_3d_level_over_the_whole_float_range(dwt, oracle, m, mode):
    """One 3-D level: out of place (the fused x+y+z kernel where the volume is wide enough), in place
    (two passes), in place through the one-pass level over the halo snapshot; forward and inverse."""
    v = float_range_input(m)
    nz, ny, nx = v.shape
    want_f, want_i = expected(oracle, m, v)
    src = dwt.lib.dwt_hip_malloc(v.nbytes)
    dst = dwt.lib.dwt_hip_malloc(v.nbytes)
    assert src and dst and dwt.lib.dwt_hip_memcpy_h2d(src, v.ctypes.data, v.nbytes) == 0
    got = np.empty_like(v)
    try:
        if mode == "op":
            dwt.transform3d_op(src, dst, nx * 4, nx * ny * 4, nx, ny, nz, 1)
            assert dwt.lib.dwt_hip_memcpy_d2h(got.ctypes.data, dst, v.nbytes) == 0
            assert same_floats(got, want_f)
        else:
            if mode == "ip-fused":
                if nx < 256:
                    pytest.skip("the one-pass in-place level needs rows of 256 samples")
                dwt.set_option("vol_fused", 2)
                dwt.set_option("vol_tile_pairs", 8)
            dwt.transform3d(0, src, nx * 4, nx * ny * 4, nx, ny, nz, 1)
            assert dwt.lib.dwt_hip_memcpy_d2h(got.ctypes.data, src, v.nbytes) == 0
            assert same_floats(got, want_f), "forward"
            assert dwt.lib.dwt_hip_memcpy_h2d(src, want_f.ctypes.data, v.nbytes) == 0
            dwt.transform3d(1, src, nx * 4, nx * ny * 4, nx, ny, nz, 1)
            assert dwt.lib.dwt_hip_memcpy_d2h(got.ctypes.data, src, v.nbytes) == 0
            assert same_floats(got, want_i), "inverse"
    finally:
        dwt.set_option("vol_fused", 1)
        dwt.set_option("vol_tile_pairs", 0)
        dwt.lib.dwt_hip_free(src)
        dwt.lib.dwt_hip_free(dst)


@pytest.mark.parametrize("klass,nf", [("subnormal", 0), ("huge", 0), ("mixed", 1)], ids=lambda v: str(v))
def test_large_image_seams_over_the_whole_float_range(dwt, oracle, klass, nf):
    """2048 x 1536, 4 levels, device resident: several tile columns and rows per level, specials on
    the seams; float 9/7 forward / inverse and the interleaved layout with its exact border strips."""
    h, w = 1536, 2048
    a = full_range_floats(np.random.default_rng(4242), (h, w), np.float32, klass, bool(nf))
    for ff, fi, f, i in (("cdf97_2f_s", "cdf97_2i_s", dwt.dwt_cdf97_2f_s, dwt.dwt_cdf97_2i_s),
                         ("cdf97_2f_inplace_s", "cdf97_2i_inplace_s", dwt.dwt_cdf97_2f_inplace_s, dwt.dwt_cdf97_2i_inplace_s)):
        want = a.copy()
        ctx = oracle.reflected_ends() if PLAIN_BUILD else warnings.catch_warnings()
        with ctx:
            oracle.fwd(ff, want, 4)
            back = want.copy()
            oracle.inv(fi, back, 4)
        d = dwt.DeviceImage(h, w).upload(a)
        f(d.ptr, d.stride_x, 4, w, h, w, h, 4)
        assert same_floats(d.download(np.float32), want), ff
        d.upload(want)
        i(d.ptr, d.stride_x, 4, w, h, w, h, 4)
        assert same_floats(d.download(np.float32), back), fi
        d.free()


# Shapes on both sides of the rules that pick a tile sweep's instantiation (dwt_sweep2d.hip, fwd_sweep_any_tile /
# inv_sweep_any_tile): the select form runs levels of 64 x 64 and more whose width is a multiple of the columns per lane
# (8 forward from 2048 columns on, else 4; groups of 4 inverse), everything else the branching form -- and a multi-level
# call crosses from one to the other on its way down.
DISPATCH_SHAPES = [(64, 64), (64, 72), (66, 64), (63, 64), (64, 60), (64, 68), (128, 66), (65, 136), (130, 2048), (96, 2056),
                   (72, 2052), (256, 126), (127, 128), (512, 520)]


@pytest.mark.parametrize("klass", ["huge", "mixed"])
@pytest.mark.parametrize("shape", DISPATCH_SHAPES, ids=lambda s: "%dx%d" % s)
def test_both_sides_of_the_instantiation_rules(dwt, oracle, shape, klass):
    """Float 9/7, float 5/3 (device-resident, 3 levels, forward and inverse) and int 5/3 over the whole range on the shapes
    above, against the faithful oracle: whichever instantiation a level takes, the line ends are the reference's."""
    from conftest import full_range_ints
    h, w = shape
    rng = np.random.default_rng(h * 4099 + w)
    a = full_range_floats(rng, (h, w), np.float32, klass, klass == "mixed")
    for name, ff, fi in (("cdf97_s", "cdf97_2f_s", "cdf97_2i_s"), ("cdf53_s", "cdf53_2f_s", "cdf53_2i_s")):
        want = a.copy()
        with (oracle.reflected_ends() if PLAIN_BUILD else warnings.catch_warnings()):
            j = oracle.fwd(ff, want, 3)
            back = want.copy()
            oracle.inv(fi, back, j)
        d = dwt.DeviceImage(h, w).upload(a)
        assert dwt.FORWARD[name](d.ptr, d.stride_x, 4, w, h, w, h, 3) == j
        assert same_floats(d.download(np.float32), want), name + " forward"
        d.upload(want)
        dwt.INVERSE[name](d.ptr, d.stride_x, 4, w, h, w, h, j)
        assert same_floats(d.download(np.float32), back), name + " inverse"
        d.free()
    b = full_range_ints(rng, (h, w))
    want = b.copy()
    j = oracle.fwd("cdf53_2f_i", want, 3)
    back = want.copy()
    oracle.inv("cdf53_2i_i", back, j)
    d = dwt.DeviceImage(h, w).upload(b)
    assert dwt.FORWARD["cdf53_i"](d.ptr, d.stride_x, 4, w, h, w, h, 3) == j
    assert np.array_equal(d.download(np.int32), want), "int 5/3 forward"
    d.upload(want)
    dwt.INVERSE["cdf53_i"](d.ptr, d.stride_x, 4, w, h, w, h, j)
    assert np.array_equal(d.download(np.int32), back), "int 5/3 inverse"
    d.free()


def test_plain_build_over_the_whole_float_range():
    """The same file against libdwt_hip_plain.so where it has been built (`make -C libdwt_amd/csrc plain`, the A/B timing
    build with reflected line ends in every float kernel): there every float entry equals the oracle's reflected-ends form.
    Own process: the library is chosen at import (DWT_HIP_LIB)."""
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(os.path.dirname(here), "libdwt_amd", "libdwt_hip_plain.so")
    if PLAIN_BUILD:
        pytest.skip("already running against the plain build")
    if not os.path.exists(lib):
        pytest.skip("libdwt_amd/libdwt_hip_plain.so not built (make -C libdwt_amd/csrc plain): an A/B artefact, not shipped")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.abspath(__file__)],
                         env=dict(os.environ, DWT_HIP_LIB=lib), capture_output=True, text=True, timeout=1200, cwd=os.path.dirname(here))
    assert out.returncode == 0 and " passed" in out.stdout, (out.stdout[-2500:], out.stderr[-1500:])
