"""CPU: the oracle's interleaved-layout transforms (src/dwt-simple.c fdwt2_*, src/libdwt.c
*_inplace_s) against the fixtures generated from the reference, and against the
reference itself where it is built."""
import numpy as np
import pytest

from conftest import bits, interleaved_cases

CASES = interleaved_cases()
IDS = [m["name"] for m, _ in CASES]


def _view(m, arr):
    return arr[:, :m["size_o"][0]]


@pytest.mark.parametrize("case", CASES, ids=IDS)
@pytest.mark.parametrize("wv", ["cdf97", "cdf53"])
def test_inplace_entries_match_golden(oracle, case, wv):
    m, z = case
    buf = z["in"].copy()
    so, si = tuple(m["size_o"]), tuple(m["size_i"])
    j = oracle.fwd(f"{wv}_2f_inplace_s", _view(m, buf), m["j_in"], size_o=so, size_i=si, decompose_one=m["decompose_one"])
    assert j == m[f"{wv}.j_out"]
    assert np.array_equal(bits(buf), bits(z[f"{wv}.fwd"]))
    oracle.inv(f"{wv}_2i_inplace_s", _view(m, buf), j, size_o=so, size_i=si, decompose_one=m["decompose_one"])
    assert np.array_equal(bits(buf), bits(z[f"{wv}.inv"]))
    # the round trip restores the transformed region (abs 1e-3 is the reference's own criterion, libdwt.c:1604)
    w, h = si
    assert np.abs(buf[:h, :w] - z["in"][:h, :w]).max() < 1e-3


@pytest.mark.parametrize("case", [c for c in CASES if "cdf97.fdwt2" in c[1]], ids=[m["name"] for m, z in CASES if "cdf97.fdwt2" in z])
@pytest.mark.parametrize("wv", ["cdf97", "cdf53"])
def test_fdwt2_matches_golden(oracle, case, wv):
    m, z = case
    buf = z["in"].copy()
    j = oracle.fdwt2(wv, _view(m, buf), m["j_in"], m["decompose_one"])
    assert j == m[f"{wv}.j_out"]
    assert np.array_equal(bits(buf), bits(z[f"{wv}.fdwt2"]))


def test_fdwt2_cdf97_equals_inplace_entry_and_differs_from_plain_separable_only_in_rounding(oracle):
    """The phase order (all rows' prolog, all columns' prolog, cores, epilogs) changes fp32
    rounding in the border bands only; the interior equals rows-then-columns bit for bit."""
    rng = np.random.default_rng(3)
    a = rng.random((96, 80), dtype=np.float32)
    b, c = a.copy(), a.copy()
    oracle.fdwt2("cdf97", b, 1)
    oracle.fwd("cdf97_2f_inplace_s", c, 1)
    assert np.array_equal(bits(b), bits(c))
    # plain separable single level on the same data, interleaved: lines through the 1-D kernel
    d = a.copy()
    for y in range(d.shape[0]):
        d[y, :] = _il_line(oracle, d[y, :].copy())
    for x in range(d.shape[1]):
        d[:, x] = _il_line(oracle, d[:, x].copy())
    diff = bits(b) != bits(d)
    assert not diff[8:-8, 8:-8].any()
    assert np.abs(b - d).max() < 1e-5 * np.abs(b).max()


def _il_line(oracle, line):
    """1-D forward 9/7 of the Mallat path before its de-interleave (even = L, odd = H)."""
    return oracle.line("cdf97_f_s", line)


@pytest.mark.parametrize("shape", [(8, 8), (37, 53), (5, 64), (64, 5), (1, 17), (17, 1), (3, 3), (4, 12), (130, 67)])
@pytest.mark.parametrize("d1", [0, 1])
def test_oracle_bitwise_equals_reference_interleaved(oracle, reference, shape, d1):
    reference.lib.dwt_util_set_num_workers(1)
    h, w = shape
    rng = np.random.default_rng(h * 131 + w)
    a = rng.random((h, w), dtype=np.float32)
    for wv in ("cdf97", "cdf53"):
        for sched in ("horizontal", "vertical", "diagonal"):
            b, c = a.copy(), a.copy()
            assert reference.fdwt2(wv, b, -1, d1, sched) == oracle.fdwt2(wv, c, -1, d1)
            assert np.array_equal(bits(b), bits(c)), (wv, sched)
        b, c = a.copy(), a.copy()
        jr = reference.fwd(f"{wv}_2f_inplace_s", b, -1, decompose_one=d1)
        assert jr == oracle.fwd(f"{wv}_2f_inplace_s", c, -1, decompose_one=d1)
        assert np.array_equal(bits(b), bits(c)), wv
        reference.inv(f"{wv}_2i_inplace_s", b, jr, decompose_one=d1)
        oracle.inv(f"{wv}_2i_inplace_s", c, jr, decompose_one=d1)
        assert np.array_equal(bits(b), bits(c)), wv


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 9, 37, 100, 257])
def test_one_dimensional_entries_equal_the_one_row_image(oracle, reference, n):
    """fdwt1_cdf97_horizontal_s / fdwt1_single_cdf97_horizontal_s (src/dwt-simple.c:2059, 2118) are the
    2-D driver on a one-row image with decompose_one: what the device entries rely on."""
    import ctypes as C

    R, O = reference.lib, oracle.lib
    R.fdwt1_cdf97_horizontal_s.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    R.fdwt1_single_cdf97_horizontal_s.argtypes = [C.c_void_p, C.c_int, C.c_int]
    O.oracle_fdwt2_cdf97_s.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    rng = np.random.default_rng(n)
    for stride in (4, 12):
        for jreq in (-1, 1, 2, 5):
            a = rng.random(n * stride // 4 + 4, dtype=np.float32)
            b, c = a.copy(), a.copy()
            j1, j2 = C.c_int(jreq), C.c_int(jreq)
            R.fdwt1_cdf97_horizontal_s(b.ctypes.data, n, stride, C.byref(j1))
            O.oracle_fdwt2_cdf97_s(c.ctypes.data, n, 1, stride * n + 64, stride, C.byref(j2), 1)
            assert j1.value == j2.value and np.array_equal(bits(b), bits(c))
        a = rng.random(n * 3 + 4, dtype=np.float32)
        b, c = a.copy(), a.copy()
        R.fdwt1_single_cdf97_horizontal_s(b.ctypes.data, n, 12)
        j2 = C.c_int(1)
        O.oracle_fdwt2_cdf97_s(c.ctypes.data, n, 1, 12 * n + 64, 12, C.byref(j2), 1)
        assert np.array_equal(bits(b), bits(c))


@pytest.mark.parametrize("shape", [(8, 8), (37, 53), (5, 64), (1, 17), (17, 1), (3, 3), (130, 67)])
@pytest.mark.parametrize("d1", [0, 1])
def test_fixed_point_int_inplace_oracle_equals_reference(oracle, reference, shape, d1):
    """src/libdwt.c:17424 / :17308, including what they do above one level (strides not scaled)."""
    h, w = shape
    rng = np.random.default_rng(h * 17 + w)
    a = rng.integers(-32768, 32768, (h, w)).astype(np.int32)
    for j in (-1, 1, 2, 4):
        b, c = a.copy(), a.copy()
        jr = reference.fwd("cdf97_2f_inplace_i", b, j, decompose_one=d1)
        assert jr == oracle.fwd("cdf97_2f_inplace_i", c, j, decompose_one=d1)
        assert np.array_equal(b, c)
        reference.inv("cdf97_2i_inplace_i", b, jr, decompose_one=d1)
        oracle.inv("cdf97_2i_inplace_i", c, jr, decompose_one=d1)
        assert np.array_equal(b, c) and np.array_equal(c, a)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_fixed_point_int_inplace_matches_golden(oracle, case):
    m, z = case
    buf = z["cdf97i.in"].copy()
    so, si = tuple(m["size_o"]), tuple(m["size_i"])
    j = oracle.fwd("cdf97_2f_inplace_i", _view(m, buf), m["j_in"], size_o=so, size_i=si, decompose_one=m["decompose_one"])
    assert j == m["cdf97i.j_out"]
    assert np.array_equal(buf, z["cdf97i.fwd"])
    oracle.inv("cdf97_2i_inplace_i", _view(m, buf), j, size_o=so, size_i=si, decompose_one=m["decompose_one"])
    assert np.array_equal(buf, z["cdf97i.inv"]) and np.array_equal(buf, z["cdf97i.in"])


@pytest.mark.parametrize("shape", [(8, 8), (37, 53), (5, 64), (1, 17), (17, 1), (3, 3), (130, 67)])
def test_one_direction_drivers_equal_reference(oracle, reference, shape):
    """fdwt2h1_cdf97_vertical_s / fdwt2v1_cdf97_vertical_s (src/dwt-simple.c:1747, 1837)."""
    import ctypes as C

    sig = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    h, w = shape
    rng = np.random.default_rng(h * 19 + w)
    for which in ("h1", "v1"):
        rf = getattr(reference.lib, f"fdwt2{which}_cdf97_vertical_s")
        of = getattr(oracle.lib, f"oracle_fdwt2{which}_cdf97_s")
        rf.argtypes = of.argtypes = sig
        rf.restype = of.restype = None
        for j in (-1, 1, 2, 4):
            for d1 in (0, 1):
                a = rng.random((h, w), dtype=np.float32)
                b, c = a.copy(), a.copy()
                j1, j2 = C.c_int(j), C.c_int(j)
                rf(b.ctypes.data, w, h, b.strides[0], 4, C.byref(j1), d1)
                of(c.ctypes.data, w, h, c.strides[0], 4, C.byref(j2), d1)
                assert j1.value == j2.value and np.array_equal(bits(b), bits(c)), (which, j, d1)
