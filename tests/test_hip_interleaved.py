"""GPU parity tests of the interleaved (in-place lifting) layout entries, all through the
C-ABI: libdwt.h dwt_cdf{97,53}_2{f,i}_inplace_s and dwt-simple.h fdwt2_cdf{97,53}_*.

Bars (written here on purpose):
 * 5/3 `_inplace_` pair: the reference finishes rows before columns -> BIT-EXACT.
 * 9/7 pair and fdwt2_*: the reference interleaves row and column work in phases (rows'
   prolog, columns' prolog, cores, epilogs).  The fused device sweep finishes the rows first,
   which rounds differently only in the top 8 rows and the last 5 columns of a level; those
   two strips are recomputed in the reference's order (il_exact_strips) -> BIT-EXACT too.
 * dwt_util_set_accel(1) runs the reference's phase order pass by pass over the whole image:
   BIT-EXACT for every entry (the cross-check of the strips)."""
import numpy as np
import pytest

from conftest import bits, interleaved_cases

pytestmark = pytest.mark.gpu

CASES = interleaved_cases()
IDS = [m["name"] for m, _ in CASES]
TOL = 1e-5


@pytest.fixture(scope="module")
def dwt():
    import libdwt_amd as d

    d.dwt_util_init()
    yield d
    d.dwt_util_set_accel(0)
    d.dwt_util_finish()


def close(a, b, what):
    scale = max(1e-30, float(np.abs(b).max()))
    err = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())
    assert err <= TOL * scale, f"{what}: max abs diff {err:.3e} vs scale {scale:.3e}"


def check(got, want, exact, what):
    if exact:
        assert np.array_equal(bits(got), bits(want)), what + " differs from the reference bit pattern"
    else:
        close(got, want, what)


@pytest.mark.parametrize("accel", [0, 1])
@pytest.mark.parametrize("wv", ["cdf97", "cdf53"])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_golden_inplace_entries_host(dwt, case, wv, accel):
    m, z = case
    dwt.dwt_util_set_accel(accel)
    try:
        buf = z["in"].copy()
        (sox, soy), (six, siy) = m["size_o"], m["size_i"]
        fwd = getattr(dwt, f"dwt_{wv}_2f_inplace_s")
        inv = getattr(dwt, f"dwt_{wv}_2i_inplace_s")
        j = fwd(buf, buf.strides[0], 4, sox, soy, six, siy, m["j_in"], m["decompose_one"])
        assert j == m[f"{wv}.j_out"]
        exact = True  # every entry, fused path (accel 0) and phase passes (accel 1) alike
        check(buf, z[f"{wv}.fwd"], exact, "forward")
        # the inverse is checked on the reference's own coefficients
        buf = z[f"{wv}.fwd"].copy()
        inv(buf, buf.strides[0], 4, sox, soy, six, siy, j, m["decompose_one"])
        check(buf, z[f"{wv}.inv"], exact, "inverse")
    finally:
        dwt.dwt_util_set_accel(0)


@pytest.mark.parametrize("accel", [0, 1])
@pytest.mark.parametrize("sched", ["horizontal", "vertical", "diagonal"])
@pytest.mark.parametrize("wv", ["cdf97", "cdf53"])
@pytest.mark.parametrize("case", [c for c in CASES if "cdf97.fdwt2" in c[1]], ids=[m["name"] for m, z in CASES if "cdf97.fdwt2" in z])
def test_golden_fdwt2_host(dwt, case, wv, sched, accel):
    m, z = case
    buf = z["in"].copy()
    w, h = m["size_o"]
    dwt.dwt_util_set_accel(accel)
    try:
        j = getattr(dwt, f"fdwt2_{wv}_{sched}_s")(buf, w, h, buf.strides[0], 4, m["j_in"], m["decompose_one"])
    finally:
        dwt.dwt_util_set_accel(0)
    assert j == m[f"{wv}.j_out"]
    check(buf, z[f"{wv}.fdwt2"], True, "fdwt2")
    # pitch padding untouched
    assert np.array_equal(bits(buf[:, w:]), bits(z["in"][:, w:]))


SHAPES = [(512, 512), (1000, 1000), (300, 513), (64, 2048), (2050, 130), (5, 7), (2, 2), (1536, 2048), (1, 300), (300, 1)]


def to_device(dwt, a, pitch):
    h, w = a.shape
    buf = np.zeros((h, pitch // 4), np.float32)
    buf[:, :w] = a
    return dwt.DeviceImage(h, w, 4, pitch).upload(buf)


def from_device(img):
    return img.download(np.float32)[:, :img.w]


def _oracle_fwd(oracle, kind, a, j, d1=0):
    b = a.copy()
    if kind == "fdwt2_cdf97":
        return b, oracle.fdwt2("cdf97", b, j, d1)
    if kind == "fdwt2_cdf53":
        return b, oracle.fdwt2("cdf53", b, j, d1)
    jj = oracle.fwd(f"{kind}_2f_inplace_s", b, j, decompose_one=d1)
    return b, jj


@pytest.mark.parametrize("kind", ["cdf97", "cdf53", "fdwt2_cdf97", "fdwt2_cdf53"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("inplace", [True, False], ids=["inplace", "outofplace"])
def test_device_resident_vs_oracle(dwt, oracle, kind, shape, inplace):
    h, w = shape
    rng = np.random.default_rng(h * 7 + w)
    a = (rng.random((h, w), dtype=np.float32) * 2 - 1)
    d1 = 1 if min(h, w) == 1 else 0
    want, jw = _oracle_fwd(oracle, kind, a, -1, d1)
    flavour = 1 if kind.startswith("fdwt2") else 0
    wname = "cdf97_s" if kind.endswith("97") else "cdf53_s"
    pitch = ((w * 4 + 255) // 256) * 256
    src = to_device(dwt, a, pitch)
    dst = src if inplace else to_device(dwt, np.full_like(a, 7.0), pitch)
    j = dwt.transform2d_interleaved(wname, 0, flavour, src.ptr, dst.ptr, pitch, 4, w, h, None, None, -1, d1)
    assert j == jw
    got = from_device(dst)
    exact = True
    check(got, want, exact, "forward")
    if not inplace:
        assert np.array_equal(bits(from_device(src)), bits(a)), "source image modified"
    if min(h, w) > 16:
        # the interior of level 0 (odd rows / columns are final there) is bit-identical in every flavour
        gi, wi = got[9:-9, 9:-9], want[9:-9, 9:-9]
        odd = np.zeros_like(gi, dtype=bool)
        odd[(np.arange(gi.shape[0]) + 9) % 2 == 1, :] = True
        odd[:, (np.arange(gi.shape[1]) + 9) % 2 == 1] = True
        assert np.array_equal(bits(gi)[odd], bits(wi)[odd])
    if flavour == 0:
        # inverse of the reference's coefficients, in place on the device
        rec_want = want.copy()
        oracle.inv(f"{kind}_2i_inplace_s", rec_want, jw, decompose_one=d1)
        src.free()
        src = to_device(dwt, want, pitch)
        if inplace:
            dst = src
        dwt.transform2d_interleaved(wname, 1, 0, src.ptr, dst.ptr, pitch, 4, w, h, None, None, jw, d1)
        check(from_device(dst), rec_want, exact, "inverse")
        close(from_device(dst), a, "round trip")
    src.free()
    if not inplace:
        dst.free()


@pytest.mark.parametrize("wv", ["cdf97", "cdf53"])
def test_sparse_frame_and_levels(dwt, oracle, wv):
    rng = np.random.default_rng(11)
    a = rng.random((200, 300), dtype=np.float32)
    for (six, siy, j, d1) in [(300, 200, 3, 0), (250, 160, -1, 0), (33, 200, 2, 1), (300, 1, -1, 1)]:
        want = a.copy()
        jw = oracle.fwd(f"{wv}_2f_inplace_s", want, j, size_o=(300, 200), size_i=(six, siy), decompose_one=d1)
        got = a.copy()
        jg = getattr(dwt, f"dwt_{wv}_2f_inplace_s")(got, got.strides[0], 4, 300, 200, six, siy, j, d1)
        assert jg == jw
        check(got, want, True, f"sparse forward {six}x{siy}")
        rec = want.copy()
        oracle.inv(f"{wv}_2i_inplace_s", rec, jw, size_o=(300, 200), size_i=(six, siy), decompose_one=d1)
        got = want.copy()
        getattr(dwt, f"dwt_{wv}_2i_inplace_s")(got, got.strides[0], 4, 300, 200, six, siy, jw, d1)
        check(got, rec, True, f"sparse inverse {six}x{siy}")
        # the exact phase-ordered path
        dwt.dwt_util_set_accel(1)
        try:
            got = a.copy()
            getattr(dwt, f"dwt_{wv}_2f_inplace_s")(got, got.strides[0], 4, 300, 200, six, siy, j, d1)
            check(got, want, True, f"sparse forward {six}x{siy}, accel 1")
            got = want.copy()
            getattr(dwt, f"dwt_{wv}_2i_inplace_s")(got, got.strides[0], 4, 300, 200, six, siy, jw, d1)
            check(got, rec, True, f"sparse inverse {six}x{siy}, accel 1")
        finally:
            dwt.dwt_util_set_accel(0)


@pytest.mark.parametrize("wv", ["cdf97", "cdf53"])
@pytest.mark.parametrize("inplace", [True, False], ids=["inplace", "outofplace"])
def test_device_sparse_frame(dwt, oracle, wv, inplace):
    """size_i < size_o on device-resident images: only the inner region is transformed, the rest
    of the outer frame keeps (in place) or receives (out of place) the source's values."""
    h, w, six, siy = 700, 900, 650, 333
    rng = np.random.default_rng(5)
    a = rng.random((h, w), dtype=np.float32)
    want = a.copy()
    jw = oracle.fwd(f"{wv}_2f_inplace_s", want, 4, size_o=(w, h), size_i=(six, siy))
    pitch = 3840
    src = to_device(dwt, a, pitch)
    dst = src if inplace else to_device(dwt, np.full_like(a, -3.0), pitch)
    j = dwt.transform2d_interleaved(f"{wv}_s", 0, 0, src.ptr, dst.ptr, pitch, 4, w, h, six, siy, 4)
    assert j == jw
    check(from_device(dst), want, True, "sparse device forward")
    rec = want.copy()
    oracle.inv(f"{wv}_2i_inplace_s", rec, jw, size_o=(w, h), size_i=(six, siy))
    src2 = to_device(dwt, want, pitch)
    dst2 = src2 if inplace else to_device(dwt, np.full_like(a, -3.0), pitch)
    dwt.transform2d_interleaved(f"{wv}_s", 1, 0, src2.ptr, dst2.ptr, pitch, 4, w, h, six, siy, jw)
    check(from_device(dst2), rec, True, "sparse device inverse")
    for d in {src, dst, src2, dst2}:
        d.free()


def test_full_size_round_trip_and_linearity(dwt):
    """8192^2, 5 levels, device resident: properties that need no CPU transform of the image."""
    n, J = 8192, 5
    rng = np.random.default_rng(42)
    a = rng.random((n, n), dtype=np.float32)
    b = rng.random((n, n), dtype=np.float32)
    out = {}
    src = dwt.DeviceImage(n, n)
    dst = dwt.DeviceImage(n, n)
    for name, t in (("a", a), ("b", b), ("ab", a + 2 * b)):
        src.upload(t)
        assert dwt.transform2d_interleaved("cdf97_s", 0, 0, src.ptr, dst.ptr, n * 4, 4, n, n, None, None, J) == J
        out[name] = dst.download(np.float32)
    lin = np.abs(out["ab"] - (out["a"] + 2 * out["b"])).max()
    assert lin <= 1e-4 * np.abs(out["ab"]).max()
    # in-place inverse of the coefficients of a
    src.upload(out["a"])
    dwt.transform2d_interleaved("cdf97_s", 1, 0, src.ptr, src.ptr, n * 4, 4, n, n, None, None, J)
    assert np.abs(src.download(np.float32) - a).max() < 1e-4
    # away from the top rows and the last columns (where the interleaved entries follow the
    # reference's phase order, a different rounding order than the Mallat drivers' rows-then-
    # columns) the level-0 detail samples equal the Mallat transform's bit for bit
    src.upload(a)
    dwt.transform2d_batch("cdf97_s", 0, src.ptr, dst.ptr, n * n * 4, 1, n * 4, n, n, J)
    mal = dst.download(np.float32)
    h = n // 2
    il = out["a"]
    assert np.array_equal(bits(il[1::2, 1::2][4:, :-3]), bits(mal[h:, h:][4:, :-3]))  # HH
    assert np.array_equal(bits(il[0::2, 1::2][4:, :-3]), bits(mal[:h, h:][4:, :-3]))  # HL
    assert np.array_equal(bits(il[1::2, 0::2][4:, :-3]), bits(mal[h:, :h][4:, :-3]))  # LH
    assert not np.array_equal(bits(il[1::2, 1::2][:4]), bits(mal[h:, h:][:4])), "the phase-ordered strips should differ in rounding"
    # and the deepest LL band sits on the stride-2^J lattice (same values within rounding)
    q = n >> J
    ll_il, ll_mal = il[:: 1 << J, :: 1 << J], mal[:q, :q]
    assert np.abs(ll_il - ll_mal).max() <= 1e-5 * np.abs(ll_mal).max()
    assert np.array_equal(bits(ll_il[8:, :-8]), bits(ll_mal[8:, :-8]))
    src.free()
    dst.free()


def test_errors(dwt):
    a = np.zeros((8, 8), np.float32)
    with pytest.raises(dwt.DwtError):
        dwt.transform2d_interleaved("cdf53_i", 0, 0, a, a, 32, 4, 8, 8)
    with pytest.raises(dwt.DwtError):
        dwt.transform2d_interleaved("cdf97_s", 1, 1, a, a, 32, 4, 8, 8)
    with pytest.raises(dwt.DwtError):
        dwt.transform2d_interleaved("cdf97_s", 0, 0, a, a, 32, 4, 8, 8, 9, 8)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8, 37, 100, 1001, 4096])
@pytest.mark.parametrize("stride", [4, 12])
def test_one_dimensional_entries(dwt, oracle, n, stride):
    """dwt-simple.h's complete 1-D transforms (fdwt1_*): a strided line through the device path of a
    one-row image; bit-identical to the reference (oracle pinned in tests/test_oracle_interleaved.py)."""
    import ctypes as C

    L = dwt.lib
    L.fdwt1_cdf97_horizontal_s.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.fdwt1_single_cdf97_horizontal_s.argtypes = [C.c_void_p, C.c_int, C.c_int]
    rng = np.random.default_rng(n * 5 + stride)
    for jreq in (-1, 1, 3):
        a = rng.random(n * stride // 4 + 4, dtype=np.float32)
        got, want = a.copy(), a.copy()
        jg = C.c_int(jreq)
        L.fdwt1_cdf97_horizontal_s(got.ctypes.data, n, stride, C.byref(jg))
        view = np.lib.stride_tricks.as_strided(want, shape=(1, n), strides=(stride * n + 64, stride))
        jw = C.c_int(jreq)
        oracle.lib.oracle_fdwt2_cdf97_s.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
        oracle.lib.oracle_fdwt2_cdf97_s(want.ctypes.data, n, 1, stride * n + 64, stride, C.byref(jw), 1)
        assert jg.value == jw.value
        assert np.array_equal(bits(got), bits(want)), (n, stride, jreq)
    a = rng.random(n * stride // 4 + 4, dtype=np.float32)
    got, want = a.copy(), a.copy()
    L.fdwt1_single_cdf97_horizontal_s(got.ctypes.data, n, stride)
    jw = C.c_int(1)
    oracle.lib.oracle_fdwt2_cdf97_s(want.ctypes.data, n, 1, stride * n + 64, stride, C.byref(jw), 1)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("shape", [(8, 8), (37, 53), (64, 5), (1, 17), (17, 1), (100, 100), (256, 192), (2, 2), (700, 900)],
                         ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("j", [-1, 1, 3])
def test_fixed_point_int_inplace_entries(dwt, oracle, shape, j):
    """dwt_cdf97_2f_inplace_i / dwt_cdf97_2i_inplace_i (src/libdwt.c:17424, 17308): bit-exact, the
    reference's unscaled-stride multi-level behaviour included; host and device pointers."""
    h, w = shape
    rng = np.random.default_rng(h * 13 + w + j)
    a = rng.integers(-32768, 32768, (h, w)).astype(np.int32)
    d1 = 1 if min(h, w) == 1 else 0
    want = a.copy()
    jw = oracle.fwd("cdf97_2f_inplace_i", want, j, decompose_one=d1)
    got = a.copy()
    assert dwt.dwt_cdf97_2f_inplace_i(got, got.strides[0], 4, w, h, w, h, j, d1) == jw
    assert np.array_equal(got, want)
    dwt.dwt_cdf97_2i_inplace_i(got, got.strides[0], 4, w, h, w, h, jw, d1)
    assert np.array_equal(got, a), "exact round trip"
    # device resident, out of place
    pitch = ((w * 4 + 63) // 64) * 64
    pad = np.zeros((h, pitch // 4), np.int32)
    pad[:, :w] = a
    src = dwt.DeviceImage(h, w, 4, pitch).upload(pad)
    dst = dwt.DeviceImage(h, w, 4, pitch).upload(np.full_like(pad, 5))
    assert dwt.transform2d_interleaved("cdf97_i", 0, 0, src.ptr, dst.ptr, pitch, 4, w, h, None, None, j, d1) == jw
    assert np.array_equal(dst.download(np.int32)[:, :w], want)
    src.free()
    dst.free()


@pytest.mark.parametrize("shape", [(2, 2), (5, 4), (37, 53), (64, 65), (300, 513)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_fixed_point_int_inplace_entries_over_the_whole_int32_range(dwt, oracle, shape):
    """dwt_cdf97_2{f,i}_inplace_i wrap modulo 2^32 like the compiled reference: samples over the whole
    int32 range, host and device pointers."""
    from conftest import full_range_ints
    h, w = shape
    a = full_range_ints(np.random.default_rng(h * 17 + w), (h, w))
    for j in (1, -1):
        want = a.copy()
        jw = oracle.fwd("cdf97_2f_inplace_i", want, j)
        rec = want.copy()
        oracle.inv("cdf97_2i_inplace_i", rec, jw)
        got = a.copy()
        assert dwt.dwt_cdf97_2f_inplace_i(got, got.strides[0], 4, w, h, w, h, j, 0) == jw
        assert np.array_equal(got, want)
        dwt.dwt_cdf97_2i_inplace_i(got, got.strides[0], 4, w, h, w, h, jw, 0)
        assert np.array_equal(got, rec)
        pitch = ((w * 4 + 63) // 64) * 64
        pad = np.zeros((h, pitch // 4), np.int32)
        pad[:, :w] = a
        src = dwt.DeviceImage(h, w, 4, pitch).upload(pad)
        dst = dwt.DeviceImage(h, w, 4, pitch).upload(np.full_like(pad, 5))
        assert dwt.transform2d_interleaved("cdf97_i", 0, 0, src.ptr, dst.ptr, pitch, 4, w, h, None, None, j, 0) == jw
        assert np.array_equal(dst.download(np.int32)[:, :w], want)
        src.free()
        dst.free()


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_golden_fixed_point_int_inplace_host(dwt, case):
    """dwt_cdf97_2f_inplace_i / _2i_inplace_i against the reference's own outputs, host-pointer entry."""
    m, z = case
    buf = z["cdf97i.in"].copy()
    (sox, soy), (six, siy) = m["size_o"], m["size_i"]
    j = dwt.dwt_cdf97_2f_inplace_i(buf, buf.strides[0], 4, sox, soy, six, siy, m["j_in"], m["decompose_one"])
    assert j == m["cdf97i.j_out"]
    assert np.array_equal(buf, z["cdf97i.fwd"])
    dwt.dwt_cdf97_2i_inplace_i(buf, buf.strides[0], 4, sox, soy, six, siy, j, m["decompose_one"])
    assert np.array_equal(buf, z["cdf97i.inv"])


@pytest.mark.parametrize("shape", [(8, 8), (37, 53), (64, 5), (1, 17), (17, 1), (100, 100), (513, 700), (3, 3)], ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("which", ["h1", "v1"])
def test_one_direction_entries(dwt, oracle, shape, which):
    """fdwt2h1_cdf97_vertical_s / fdwt2v1_cdf97_vertical_s (src/dwt-simple.c:1747, 1837): rows only /
    columns only at every level; one direction has no phase interleaving, so bit-exact."""
    import ctypes as C

    h, w = shape
    sig = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    fn = getattr(dwt.lib, f"fdwt2{which}_cdf97_vertical_s")
    ofn = getattr(oracle.lib, f"oracle_fdwt2{which}_cdf97_s")
    fn.argtypes = ofn.argtypes = sig
    fn.restype = ofn.restype = None
    rng = np.random.default_rng(h + 3 * w)
    for j in (-1, 1, 3):
        for d1 in (0, 1):
            a = rng.random((h, w + 2), dtype=np.float32)
            got, want = a.copy(), a.copy()
            jg, jw = C.c_int(j), C.c_int(j)
            fn(got.ctypes.data, w, h, got.strides[0], 4, C.byref(jg), d1)
            ofn(want.ctypes.data, w, h, want.strides[0], 4, C.byref(jw), d1)
            assert jg.value == jw.value
            assert np.array_equal(bits(got), bits(want)), (which, shape, j, d1)


@pytest.mark.parametrize("shape,levels", [((2048, 2048), 5), ((1500, 1000), 3), ((3001, 4097), 5), ((4096, 1024), 4), ((640, 2000), 2),
                                          ((8192, 8192), 5)], ids=lambda v: str(v))
def test_border_strips_in_the_sweeps_launch(dwt, oracle, shape, levels):
    """9/7, device resident, out of place, both directions: the samples whose rounding depends on the reference's
    phase order (rows 0..7, the last 8 columns of every level) are computed by extra workgroups of each level's own
    sweep launch while its tiles leave them alone.  Same bits as the reference's order pass by pass over the whole
    image (accel 1), and as the oracle where the oracle is quick enough."""
    h, w = shape
    rng = np.random.default_rng(h * 3 + w + levels)
    img = rng.random((h, w), dtype=np.float32)
    src = dwt.DeviceImage(h, w).upload(img)
    outs = []
    for accel in (0, 1):
        dst = dwt.DeviceImage(h, w).upload(np.full((h, w), -9.0, np.float32))
        back = dwt.DeviceImage(h, w).upload(np.full((h, w), -7.0, np.float32))
        dwt.dwt_util_set_accel(accel)
        try:
            j = dwt.transform2d_interleaved("cdf97_s", 0, 0, src.ptr, dst.ptr, w * 4, 4, w, h, None, None, levels)
            dwt.transform2d_interleaved("cdf97_s", 1, 0, dst.ptr, back.ptr, w * 4, 4, w, h, None, None, levels)
        finally:
            dwt.dwt_util_set_accel(0)
        assert j == levels
        outs.append((dst.download(np.float32), back.download(np.float32)))
        dst.free()
        back.free()
    assert np.array_equal(bits(outs[0][0]), bits(outs[1][0])), "forward"
    assert np.array_equal(bits(outs[0][1]), bits(outs[1][1])), "inverse"
    if h * w <= 3001 * 4097:
        want = img.copy()
        oracle.fwd("cdf97_2f_inplace_s", want, levels)
        assert np.array_equal(bits(outs[0][0]), bits(want))
        oracle.inv("cdf97_2i_inplace_s", want, levels)
        assert np.array_equal(bits(outs[0][1]), bits(want))
    src.free()


@pytest.mark.parametrize("shape,levels", [((8192, 8192), 5), ((4096, 4096), 1), ((3000, 5000), 3), ((6000, 2056), 2), ((2500, 2311), 4), ((4100, 2568), 2)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("kind", ["cdf97", "cdf53"])
def test_in_place_level_over_the_halo_snapshot(dwt, kind, shape, levels):
    """In-place calls on images of more than 4 M samples: level 0 runs IN PLACE over a snapshot of what its tiles read
    of their neighbours (and of what the border strip waves of the same launch write) instead of through a staging
    copy of the image.  Same bits as the out-of-place call and as the staging path (il_inplace_shell = 0), forward
    and inverse, several times over (a missed dependency would show up as a race)."""
    h, w = shape
    rng = np.random.default_rng(h + 7 * w + levels)
    img = rng.random((h, w), dtype=np.float32)
    src = dwt.DeviceImage(h, w).upload(img)
    ref_f = dwt.DeviceImage(h, w).upload(np.zeros((h, w), np.float32))
    ref_i = dwt.DeviceImage(h, w).upload(np.zeros((h, w), np.float32))
    name = kind + "_s"
    dwt.transform2d_interleaved(name, 0, 0, src.ptr, ref_f.ptr, w * 4, 4, w, h, None, None, levels)
    dwt.transform2d_interleaved(name, 1, 0, ref_f.ptr, ref_i.ptr, w * 4, 4, w, h, None, None, levels)
    want_f, want_i = ref_f.download(np.float32), ref_i.download(np.float32)
    work = dwt.DeviceImage(h, w)
    for shell in (1, 1, 0, 1):
        dwt.set_option("il_inplace_shell", shell)
        try:
            work.upload(img)
            dwt.transform2d_interleaved(name, 0, 0, work.ptr, work.ptr, w * 4, 4, w, h, None, None, levels)
            got_f = work.download(np.float32)
            dwt.transform2d_interleaved(name, 1, 0, work.ptr, work.ptr, w * 4, 4, w, h, None, None, levels)
            got_i = work.download(np.float32)
        finally:
            dwt.set_option("il_inplace_shell", 1)
        assert np.array_equal(bits(got_f), bits(want_f)), ("forward", shell)
        assert np.array_equal(bits(got_i), bits(want_i)), ("inverse", shell)
    for d in (src, ref_f, ref_i, work):
        d.free()


@pytest.mark.parametrize("shape,levels", [((1500, 1000), 3), ((2048, 2048), 5), ((640, 2000), 2)], ids=lambda v: str(v))
def test_fast_borders_option_within_tolerance(dwt, oracle, shape, levels):
    """Option il_exact_borders = 0 (opt-in, like "fma"): the interleaved 9/7 without the exact border strips --
    the top rows / last columns of every level keep the fused sweep's rows-then-columns rounding instead of the
    reference's phase order.  NOT the reference's bits there, but within the north star's 1e-5 relative
    tolerance everywhere, identical away from the borders, and the inverse restores the image; the default
    stays bit-exact (every other test of this file)."""
    h, w = shape
    img = np.random.default_rng(h + 7 * w).random((h, w), dtype=np.float32)
    want = img.copy()
    jw = oracle.fwd("cdf97_2f_inplace_s", want, levels)
    src = dwt.DeviceImage(h, w).upload(img)
    dst = dwt.DeviceImage(h, w).upload(np.zeros_like(img))
    dwt.set_option("il_exact_borders", 0)
    try:
        assert dwt.get_option("il_exact_borders") == 0
        assert dwt.transform2d_interleaved("cdf97_s", 0, 0, src.ptr, dst.ptr, w * 4, 4, w, h, None, None, levels) == jw
        got = dst.download(np.float32)
        assert float(np.abs(got - want).max() / np.abs(want).max()) <= 1e-5
        # away from the image's top / right border region (8 rows, 8 columns of the deepest level) nothing differs
        reach = 8 << (levels - 1)
        assert np.array_equal(bits(got[reach:, :w - reach]), bits(want[reach:, :w - reach]))
        assert not np.array_equal(bits(got), bits(want)), "the option should have skipped the strips"
        dwt.transform2d_interleaved("cdf97_s", 1, 0, dst.ptr, src.ptr, w * 4, 4, w, h, None, None, levels)
        assert np.abs(src.download(np.float32) - img).max() < 1e-4
    finally:
        dwt.set_option("il_exact_borders", 1)
    src.free()
    dst.free()


@pytest.mark.parametrize("opts", [{"waves": 1}, {"waves": 2}, {"ring": 16}, {"tile_pairs": 8}, {"tile_pairs": 16, "waves": 3}, {"tile_pairs": 64, "ring": 16},
                                  {"xcd_swizzle": 0}], ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
def test_interleaved_bits_do_not_depend_on_the_sweep_settings(dwt, opts):
    """The strip waves share their launch with the tile waves (workgroups of `waves` waves), the in-place snapshot is cut
    for the tile height in force: whatever the sweep settings, 9/7 interleaved calls -- out of place and in place, forward
    and inverse -- give the bits of the reference's phase order pass by pass (accel 1)."""
    h, w, levels = 2200, 2312, 3
    rng = np.random.default_rng(17)
    img = rng.random((h, w), dtype=np.float32)
    src = dwt.DeviceImage(h, w).upload(img)
    f = dwt.DeviceImage(h, w); r = dwt.DeviceImage(h, w); ip = dwt.DeviceImage(h, w)
    dwt.dwt_util_set_accel(1)
    try:
        dwt.transform2d_interleaved("cdf97_s", 0, 0, src.ptr, f.ptr, w * 4, 4, w, h, None, None, levels)
        dwt.transform2d_interleaved("cdf97_s", 1, 0, f.ptr, r.ptr, w * 4, 4, w, h, None, None, levels)
    finally:
        dwt.dwt_util_set_accel(0)
    want_f, want_r = f.download(np.float32), r.download(np.float32)
    try:
        for k, v in opts.items():
            dwt.set_option(k, v)
        dwt.transform2d_interleaved("cdf97_s", 0, 0, src.ptr, f.ptr, w * 4, 4, w, h, None, None, levels)
        dwt.transform2d_interleaved("cdf97_s", 1, 0, f.ptr, r.ptr, w * 4, 4, w, h, None, None, levels)
        got_f, got_r = f.download(np.float32), r.download(np.float32)
        ip.upload(img)
        dwt.transform2d_interleaved("cdf97_s", 0, 0, ip.ptr, ip.ptr, w * 4, 4, w, h, None, None, levels)
        got_fi = ip.download(np.float32)
        dwt.transform2d_interleaved("cdf97_s", 1, 0, ip.ptr, ip.ptr, w * 4, 4, w, h, None, None, levels)
        got_ri = ip.download(np.float32)
    finally:
        for k in opts:
            dwt.set_option(k, {"waves": 4, "ring": 0, "tile_pairs": 0, "xcd_swizzle": 1}[k])
    for got, want, what in ((got_f, want_f, "forward"), (got_fi, want_f, "forward in place"), (got_r, want_r, "inverse"), (got_ri, want_r, "inverse in place")):
        assert np.array_equal(bits(got), bits(want)), what
    for d in (src, f, r, ip):
        d.free()
