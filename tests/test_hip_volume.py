"""GPU parity of the 3-D path (BASELINE config 5): single level against the golden
vectors of cdf97_3f_ip_sep_horizontal_s / cdf97_3i_ip_sep_horizontal_s, multi-level
against the oracle applied on the LLL lattice (strides doubled per level)."""
import numpy as np
import pytest

from conftest import bits, golden_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dwt():
    import libdwt_amd as d

    d.dwt_util_init()
    yield d
    d.dwt_util_finish()


class DevVol:
    def __init__(self, dwt, arr):
        self.dwt = dwt
        self.shape = arr.shape
        self.nbytes = arr.nbytes
        self.ptr = dwt.lib.dwt_hip_malloc(arr.nbytes)
        assert self.ptr
        a = np.ascontiguousarray(arr)
        assert dwt.lib.dwt_hip_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes) == 0

    def get(self):
        out = np.empty(self.shape, np.float32)
        assert self.dwt.lib.dwt_hip_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes) == 0
        return out

    def run(self, inverse, levels):
        nz, ny, nx = self.shape
        self.dwt.transform3d(inverse, self.ptr, nx * 4, nx * ny * 4, nx, ny, nz, levels)

    def free(self):
        self.dwt.lib.dwt_hip_free(self.ptr)


def oracle_multilevel(oracle, v, levels, inverse):
    order = range(levels - 1, -1, -1) if inverse else range(levels)
    for j in order:
        s = 1 << j
        view = v[::s, ::s, ::s]
        oracle.vol("cdf97_3i_s" if inverse else "cdf97_3f_s", view)
    return v


@pytest.mark.parametrize("pad_y,pad_z", [(0, 0), (16, 0), (0, 2), (48, 3)], ids=["dense", "padded-rows", "padded-slices", "padded-both"])
@pytest.mark.parametrize("shape,levels", [((40, 64, 256), 1), ((33, 65, 129), 2), ((66, 70, 512), 3), ((24, 40, 300), 1)], ids=lambda v: str(v))
def test_in_place_forward_through_the_fused_levels(dwt, oracle, shape, levels, pad_y, pad_z):
    """dwt_hip_transform3d forward where the one-pass level applies (forced here by vol_fused=2 so
    that small volumes take it): the out-of-place levels into a result volume, one copy back; same
    bits as the two passes per level, the padding between rows and slices untouched."""
    nz, ny, nx = shape
    sy, sz = nx * 4 + pad_y, 0
    sz = sy * ny + pad_z * sy
    rng = np.random.default_rng(sum(shape) * 7 + levels)
    vol = rng.random(shape, dtype=np.float32)
    want = oracle_multilevel(oracle, vol.copy(), levels, False)
    raw = np.full(sz * nz, 0xA5, np.uint8)
    view = np.lib.stride_tricks.as_strided(raw.view(np.float32), shape=shape, strides=(sz, sy, 4))
    view[...] = vol
    ptr = dwt.lib.dwt_hip_malloc(raw.nbytes)
    assert dwt.lib.dwt_hip_memcpy_h2d(ptr, raw.ctypes.data, raw.nbytes) == 0
    dwt.set_option("vol_fused", 2)
    try:
        dwt.transform3d(0, ptr, sy, sz, nx, ny, nz, levels)
    finally:
        dwt.set_option("vol_fused", 1)
    back = np.empty_like(raw)
    assert dwt.lib.dwt_hip_memcpy_d2h(back.ctypes.data, ptr, raw.nbytes) == 0
    got = np.lib.stride_tricks.as_strided(back.view(np.float32), shape=shape, strides=(sz, sy, 4))
    assert np.array_equal(bits(np.ascontiguousarray(got)), bits(want))
    # nothing outside the samples was written
    mask = np.ones(raw.shape, bool)
    np.lib.stride_tricks.as_strided(mask, shape=(nz, ny, nx * 4), strides=(sz, sy, 1))[...] = False
    assert np.all(back[mask] == 0xA5)
    # and the two-pass in-place path agrees
    assert dwt.lib.dwt_hip_memcpy_h2d(ptr, raw.ctypes.data, raw.nbytes) == 0
    dwt.set_option("vol_inplace_fused", 0)
    try:
        dwt.transform3d(0, ptr, sy, sz, nx, ny, nz, levels)
    finally:
        dwt.set_option("vol_inplace_fused", 1)
    back2 = np.empty_like(raw)
    assert dwt.lib.dwt_hip_memcpy_d2h(back2.ctypes.data, ptr, raw.nbytes) == 0
    assert np.array_equal(back, back2)
    dwt.lib.dwt_hip_free(ptr)


def test_golden_single_level(dwt):
    for meta, src, fwd, inv in golden_cases("cdf97_3d_s"):
        d = DevVol(dwt, src)
        d.run(0, 1)
        got = d.get()
        assert np.array_equal(bits(got), bits(fwd)), meta
        d.run(1, 1)
        assert np.array_equal(bits(d.get()), bits(inv)), meta
        d.free()


@pytest.mark.parametrize("shape,levels", [((16, 16, 16), 1), ((33, 65, 129), 1), ((40, 100, 300), 1), ((64, 64, 64), 3),
                                          ((36, 52, 40), 2), ((128, 128, 512), 3), ((256, 256, 256), 3)],
                         ids=lambda v: str(v))
def test_volume_vs_oracle(dwt, oracle, shape, levels):
    rng = np.random.default_rng(sum(shape) + levels)
    vol = rng.random(shape, dtype=np.float32)
    want = oracle_multilevel(oracle, vol.copy(), levels, False)
    d = DevVol(dwt, vol)
    d.run(0, levels)
    got = d.get()
    assert np.array_equal(bits(got), bits(want))
    d.run(1, levels)
    rec = d.get()
    want_rec = oracle_multilevel(oracle, want.copy(), levels, True)
    assert np.array_equal(bits(rec), bits(want_rec))
    assert np.abs(rec - vol).max() < 1e-4
    d.free()


@pytest.mark.parametrize("fused", [1, 6, 0, -1, -2], ids=["fused", "fused-6-rows", "two-pass", "fused-dense-results", "fused-strided-stores"])
@pytest.mark.parametrize("shape,levels", [((16, 16, 256), 1), ((37, 50, 256), 1), ((9, 7, 512), 1), ((64, 96, 256), 2),
                                          ((40, 33, 768), 1), ((128, 128, 512), 3), ((66, 130, 1024), 3), ((33, 65, 129), 2),
                                          ((2, 2, 256), 1), ((130, 3, 256), 1), ((20, 40, 300), 1), ((33, 35, 129), 1),
                                          ((18, 70, 1000), 2), ((24, 24, 515), 1), ((200, 520, 600), 2), ((129, 1000, 513), 1),
                                          ((70, 66, 512), 2), ((33, 47, 1024), 4), ((129, 31, 1536), 2)],
                         ids=lambda v: str(v))
def test_out_of_place_forward_vs_oracle(dwt, oracle, shape, levels, fused):
    """dwt_hip_transform3d_op (cdf97_3f_op_sep_horizontal_s semantics): one fused x+y+z pass per
    level for volumes at least 128 samples wide (whole or overhanging 256-column tiles), two passes elsewhere; the source stays intact;
    bit-identical either way."""
    rng = np.random.default_rng(sum(shape) * 3 + levels)
    vol = rng.random(shape, dtype=np.float32)
    want = oracle_multilevel(oracle, vol.copy(), levels, False)
    src = DevVol(dwt, vol)
    dst = DevVol(dwt, np.full(shape, -5.0, np.float32))
    nz, ny, nx = shape
    # fused: 2 = the one-pass kernel wherever it can run (small test volumes included), 1 = where it pays
    dwt.set_option("vol_fused", 2 if fused else 0)
    dwt.set_option("vol_rows", 6 if fused == 6 else 8)  # output rows per wave of the fused kernel
    # levels >= 1 into their lattice of the destination: 2 (default) = level 1 writes the rows it shares
    # with level 0 whole (x sizes that are multiples of 512), 1 = strided stores, 0 = dense + scatter pass
    dwt.set_option("vol_direct", 0 if fused == -1 else 1 if fused == -2 else 2)
    # whole-tile variant of the kernel where the x size is a multiple of 256 (default) or the general one
    dwt.set_option("vol_whole", 0 if fused == -1 else 1)
    try:
        dwt.transform3d_op(src.ptr, dst.ptr, nx * 4, nx * ny * 4, nx, ny, nz, levels)
    finally:
        dwt.set_option("vol_fused", 1)
        dwt.set_option("vol_rows", 8)
        dwt.set_option("vol_direct", 2)
        dwt.set_option("vol_whole", 1)
    assert np.array_equal(bits(dst.get()), bits(want))
    assert np.array_equal(bits(src.get()), bits(vol)), "source volume modified"
    # the in-place inverse undoes it
    dst.run(1, levels)
    assert np.abs(dst.get() - vol).max() < 1e-4
    src.free()
    dst.free()


@pytest.mark.parametrize("shape,levels", [((66, 70, 512), 3), ((40, 33, 1024), 2), ((30, 50, 300), 2)], ids=lambda v: str(v))
def test_out_of_place_padded_strides(dwt, oracle, shape, levels):
    """dwt_hip_transform3d_op on volumes whose rows and slices are padded (volume_t strides): the
    fused levels, the merged rows of levels 0 / 1 and the lattice scatter all address through the
    strides; the padding of the destination stays untouched."""
    nz, ny, nx = shape
    sy = nx * 4 + 80
    sz = sy * (ny + 3)
    rng = np.random.default_rng(sum(shape) * 11 + levels)
    vol = rng.random(shape, dtype=np.float32)
    want = oracle_multilevel(oracle, vol.copy(), levels, False)
    raw = np.full(sz * nz, 0x5A, np.uint8)
    np.lib.stride_tricks.as_strided(raw.view(np.float32), shape=shape, strides=(sz, sy, 4))[...] = vol
    src = dwt.lib.dwt_hip_malloc(raw.nbytes)
    dst = dwt.lib.dwt_hip_malloc(raw.nbytes)
    fill = np.full(sz * nz, 0xC3, np.uint8)
    assert dwt.lib.dwt_hip_memcpy_h2d(src, raw.ctypes.data, raw.nbytes) == 0
    assert dwt.lib.dwt_hip_memcpy_h2d(dst, fill.ctypes.data, fill.nbytes) == 0
    dwt.set_option("vol_fused", 2)
    try:
        dwt.transform3d_op(src, dst, sy, sz, nx, ny, nz, levels)
    finally:
        dwt.set_option("vol_fused", 1)
    back = np.empty_like(raw)
    assert dwt.lib.dwt_hip_memcpy_d2h(back.ctypes.data, dst, raw.nbytes) == 0
    got = np.lib.stride_tricks.as_strided(back.view(np.float32), shape=shape, strides=(sz, sy, 4))
    assert np.array_equal(bits(np.ascontiguousarray(got)), bits(want))
    mask = np.ones(raw.shape, bool)
    np.lib.stride_tricks.as_strided(mask, shape=(nz, ny, nx * 4), strides=(sz, sy, 1))[...] = False
    assert np.all(back[mask] == 0xC3), "padding of the destination written"
    same = np.empty_like(raw)
    assert dwt.lib.dwt_hip_memcpy_d2h(same.ctypes.data, src, raw.nbytes) == 0
    assert np.array_equal(same, raw), "source modified"
    dwt.lib.dwt_hip_free(src)
    dwt.lib.dwt_hip_free(dst)


@pytest.mark.parametrize("tile_pairs", [4, 16, 64, 128])
def test_march_length_does_not_change_the_bits(dwt, oracle, tile_pairs):
    """The fused kernel's march length along z (option vol_tile_pairs; chosen by a model in
    production) only changes the schedule: z chunks of any length give the oracle's bits."""
    shape = (150, 40, 512)
    rng = np.random.default_rng(99)
    vol = rng.random(shape, dtype=np.float32)
    want = oracle_multilevel(oracle, vol.copy(), 2, False)
    src = DevVol(dwt, vol)
    dst = DevVol(dwt, np.zeros(shape, np.float32))
    nz, ny, nx = shape
    dwt.set_option("vol_fused", 2)
    dwt.set_option("vol_tile_pairs", tile_pairs)
    try:
        dwt.transform3d_op(src.ptr, dst.ptr, nx * 4, nx * ny * 4, nx, ny, nz, 2)
    finally:
        dwt.set_option("vol_fused", 1)
        dwt.set_option("vol_tile_pairs", 0)
    assert np.array_equal(bits(dst.get()), bits(want))
    src.free()
    dst.free()


def test_out_of_place_zero_levels_and_errors(dwt):
    vol = np.random.default_rng(1).random((4, 5, 6), dtype=np.float32)
    src = DevVol(dwt, vol)
    dst = DevVol(dwt, np.zeros_like(vol))
    dwt.transform3d_op(src.ptr, dst.ptr, 24, 120, 6, 5, 4, 0)
    assert np.array_equal(dst.get(), vol)
    with pytest.raises(dwt.DwtError):
        dwt.transform3d_op(src.ptr, src.ptr, 24, 120, 6, 5, 4, 1)
    src.free()
    dst.free()


IP_SHAPES = [((40, 64, 256), 1), ((33, 65, 129), 2), ((66, 70, 512), 3), ((24, 40, 300), 1), ((9, 33, 257), 1),
             ((17, 35, 260), 1), ((8, 2, 515), 1), ((41, 97, 770), 2), ((50, 31, 1030), 1), ((26, 36, 2), 1),
             ((130, 5, 64), 1), ((19, 129, 259), 1), ((64, 64, 64), 3), ((21, 34, 513), 2)]


@pytest.mark.parametrize("tile_pairs", [0, 4, 7, 16], ids=lambda v: f"march{v}")
@pytest.mark.parametrize("shape,levels", IP_SHAPES, ids=lambda v: str(v))
def test_one_pass_in_place_levels(dwt, oracle, shape, levels, tile_pairs):
    """dwt_hip_transform3d with the one-pass in-place level forced on small volumes (vol_fused = 2):
    tiles read their own part from the volume they are overwriting and their halo rows, columns and
    march-boundary slices from the shell snapshot.  Shapes cover several tile rows / columns, tiles
    that overhang the volume by 1 .. 255 columns and 1 .. 31 rows, odd depths, and marches of 4, 7 and
    16 slice pairs (so that the slice shell and the far-end reflection are exercised).  Forward and
    inverse, every level: the oracle's bits."""
    nz, ny, nx = shape
    rng = np.random.default_rng(sum(shape) * 5 + levels)
    vol = rng.random(shape, dtype=np.float32)
    want = oracle_multilevel(oracle, vol.copy(), levels, False)
    d = DevVol(dwt, vol)
    dwt.set_option("vol_fused", 2)
    dwt.set_option("vol_tile_pairs", tile_pairs)
    try:
        d.run(0, levels)
        got = d.get()
        assert np.array_equal(bits(got), bits(want)), "forward"
        d.run(1, levels)
        rec = d.get()
    finally:
        dwt.set_option("vol_fused", 1)
        dwt.set_option("vol_tile_pairs", 0)
    want_rec = oracle_multilevel(oracle, want.copy(), levels, True)
    assert np.array_equal(bits(rec), bits(want_rec)), "inverse"
    assert np.abs(rec - vol).max() < 1e-4
    d.free()


def test_placed_volumes(dwt, oracle):
    """dwt_hip_alloc_volumes: source / destination / workspace of an out-of-place 3-D call chosen by measurement
    (an arena of the card's free memory); the transform on them gives the oracle's bits; dwt_hip_free takes
    them back."""
    shape, levels = (24, 40, 256), 2
    vol = np.random.default_rng(3).random(shape, dtype=np.float32)
    want = oracle_multilevel(oracle, vol.copy(), levels, False)
    L = dwt.lib
    dwt.set_option("place_min_mib", 0)
    try:
        src, dst = dwt.alloc_volumes(shape[2], shape[1], shape[0], levels)
        rep = dwt.alloc_batch_report()
        assert rep["arena_GiB"] >= 8 and rep["dst_positions_tried"] >= 1, rep
        assert L.dwt_hip_memcpy_h2d(src, vol.ctypes.data, vol.nbytes) == 0
        dwt.transform3d_op(src, dst, shape[2] * 4, shape[2] * shape[1] * 4, shape[2], shape[1], shape[0], levels)
        got = np.empty_like(vol)
        assert L.dwt_hip_memcpy_d2h(got.ctypes.data, dst, got.nbytes) == 0
        assert np.array_equal(bits(got), bits(want))
        L.dwt_hip_free(src)
        L.dwt_hip_free(dst)
    finally:
        dwt.set_option("place_min_mib", 1024)
        dwt.dwt_util_finish()


def test_one_pass_in_place_tall_thin_volume(dwt, oracle):
    """A volume with more than 8192 / 7 tile rows of 32 (the shell row index of k_vol_level_ip fills the top
    14 bits of a packed word and is decoded unsigned): 6 x 38000 x 130, tiles of 32 rows (vol_ip_waves = 4),
    forward and inverse in one pass in place -- the oracle's bits."""
    shape = (6, 38000, 130)
    vol = np.random.default_rng(12).random(shape, dtype=np.float32)
    want = oracle_multilevel(oracle, vol.copy(), 1, False)
    d = DevVol(dwt, vol)
    dwt.set_option("vol_fused", 2)
    dwt.set_option("vol_ip_waves", 4)
    try:
        d.run(0, 1)
        assert np.array_equal(bits(d.get()), bits(want)), "forward"
        d.run(1, 1)
        rec = d.get()
    finally:
        dwt.set_option("vol_fused", 1)
        dwt.set_option("vol_ip_waves", 0)
    assert np.array_equal(bits(rec), bits(oracle_multilevel(oracle, want.copy(), 1, True))), "inverse"
    d.free()


def test_one_pass_in_place_padded_strides(dwt, oracle):
    """The in-place levels address through volume_t strides: padding between rows and slices stays
    untouched, forward and inverse."""
    shape, levels = (37, 70, 520), 2
    nz, ny, nx = shape
    sy = nx * 4 + 48
    sz = sy * (ny + 2)
    rng = np.random.default_rng(77)
    vol = rng.random(shape, dtype=np.float32)
    want = oracle_multilevel(oracle, vol.copy(), levels, False)
    raw = np.full(sz * nz, 0xA5, np.uint8)
    np.lib.stride_tricks.as_strided(raw.view(np.float32), shape=shape, strides=(sz, sy, 4))[...] = vol
    ptr = dwt.lib.dwt_hip_malloc(raw.nbytes)
    assert dwt.lib.dwt_hip_memcpy_h2d(ptr, raw.ctypes.data, raw.nbytes) == 0
    mask = np.ones(raw.shape, bool)
    np.lib.stride_tricks.as_strided(mask, shape=(nz, ny, nx * 4), strides=(sz, sy, 1))[...] = False
    dwt.set_option("vol_fused", 2)
    try:
        for inverse, ref in ((0, want), (1, oracle_multilevel(oracle, want.copy(), levels, True))):
            dwt.transform3d(inverse, ptr, sy, sz, nx, ny, nz, levels)
            back = np.empty_like(raw)
            assert dwt.lib.dwt_hip_memcpy_d2h(back.ctypes.data, ptr, raw.nbytes) == 0
            got = np.lib.stride_tricks.as_strided(back.view(np.float32), shape=shape, strides=(sz, sy, 4))
            assert np.array_equal(bits(np.ascontiguousarray(got)), bits(ref)), "inverse" if inverse else "forward"
            assert np.all(back[mask] == 0xA5), "padding written"
    finally:
        dwt.set_option("vol_fused", 1)
    dwt.lib.dwt_hip_free(ptr)


# ---- the reference's own 3-D boundary: struct volume_t, include/volume.h / volume-dwt.h ----

def _wide_cases():
    import json
    import os

    from conftest import GOLDEN

    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)["files"]["cdf97_3d_wide.npz"]["cases"]


def _sha(a):
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("fused", [2, 1, 0], ids=["one-pass", "default", "two-pass"])
@pytest.mark.parametrize("where", ["host", "device"])
@pytest.mark.parametrize("meta", _wide_cases(), ids=lambda m: m["name"])
def test_reference_fixtures_through_the_volume_api(dwt, meta, where, fused):
    """tests/golden/cdf97_3d_wide.npz -- volumes wide enough for the fused kernels, transformed by the
    REFERENCE through cdf97_3f_ip_sep_horizontal_s, cdf97_3f_op_sep_horizontal_s and
    cdf97_3i_ip_sep_horizontal_s -- replayed through the same three functions of include/volume-dwt.h on
    struct volume_t, host volumes (staged) and device volumes, with the one-pass kernels forced
    (vol_fused = 2: k_vol_fwd_fused out of place, k_vol_level_ip in place), as shipped, and in two passes.
    Inputs: the fixture's array, the seed's stream, or the product's own volume_fill_s (each checked
    against the reference's digest first).  Bit for bit."""
    import os

    from conftest import GOLDEN

    z = np.load(os.path.join(GOLDEN, "cdf97_3d_wide.npz"))
    shp = tuple(meta["shape_zyx"])
    if meta["full"]:
        v = z[meta["name"] + ".in"]
    elif meta["input"] == "rand":
        v = np.random.default_rng(meta["seed"]).random(shp, dtype=np.float32)
    else:
        v = np.zeros(shp, np.float32)
        vt = dwt.volume_of(v)
        dwt.lib.volume_fill_s(vt)
    assert _sha(v) == meta["sha256"]["in"]
    dwt.set_option("vol_fused", fused)
    try:
        if where == "host":
            ip = v.copy()
            dwt.cdf97_3f_ip_sep_horizontal_s(dwt.volume_of(ip))
            big = np.full((shp[0], shp[1] + 1, shp[2] + 5), -2.0, np.float32)  # a destination with other strides
            op = big[:, :shp[1], :shp[2]]
            src = v.copy()
            dwt.cdf97_3f_op_sep_horizontal_s(dwt.volume_of(src), dwt.volume_of(op))
            assert np.array_equal(bits(src), bits(v)), "source modified"
            assert np.all(big[:, shp[1]:, :] == -2.0) and np.all(big[:, :, shp[2]:] == -2.0), "padding written"
            inv = ip.copy()
            dwt.cdf97_3i_ip_sep_horizontal_s(dwt.volume_of(inv))
        else:
            d1, d2 = DevVol(dwt, v), DevVol(dwt, np.full(shp, -2.0, np.float32))
            strides = (shp[1] * shp[2] * 4, shp[2] * 4, 4)
            dwt.cdf97_3f_op_sep_horizontal_s(dwt.volume_of(d1.ptr, shp, strides), dwt.volume_of(d2.ptr, shp, strides))
            op = d2.get()
            assert np.array_equal(bits(d1.get()), bits(v)), "source modified"
            dwt.cdf97_3f_ip_sep_horizontal_s(dwt.volume_of(d1.ptr, shp, strides))
            ip = d1.get()
            dwt.cdf97_3i_ip_sep_horizontal_s(dwt.volume_of(d1.ptr, shp, strides))
            inv = d1.get()
            d1.free()
            d2.free()
    finally:
        dwt.set_option("vol_fused", 1)
    assert _sha(ip) == meta["sha256"]["fwd"], "in-place forward"
    assert _sha(op) == meta["sha256"]["fwd_op"], "out-of-place forward"
    assert _sha(inv) == meta["sha256"]["inv"], "inverse"
    if meta["full"]:
        assert np.array_equal(bits(ip), bits(z[meta["name"] + ".fwd"]))
        assert np.array_equal(bits(inv), bits(z[meta["name"] + ".inv"]))


def test_schedule_dispatcher(dwt):
    """cdf97_3f_op_wrapper_s (src/volume-dwt.c:2787): approaches 0..9 are schedules of the one transform
    (one kernel here, the separable schedules' bits); 10 / 11 / 12 lift x / y / z lines only -- x copies
    first, y and z work in place on the destination -- pinned by the reference's outputs."""
    import os

    from conftest import GOLDEN

    z = np.load(os.path.join(GOLDEN, "cdf97_3d_wide.npz"))
    v = z["vol_dirs.in"]
    full = v.copy()
    dwt.cdf97_3f_ip_sep_horizontal_s(dwt.volume_of(full))
    src = v.copy()  # (volume_of keeps the address only: the array has to outlive the call)
    for ap in range(10):
        dst = np.zeros_like(v)
        dwt.cdf97_3f_op_wrapper_s(dwt.volume_of(src), dwt.volume_of(dst), ap)
        assert np.array_equal(bits(dst), bits(full)), ap
    for ap, tag in ((10, "x"), (11, "y"), (12, "z")):
        dst = v.copy() if ap != 10 else np.full(v.shape, -3.0, np.float32)
        dwt.cdf97_3f_op_wrapper_s(dwt.volume_of(src), dwt.volume_of(dst), ap)
        assert np.array_equal(bits(dst), bits(z["vol_dirs." + tag])), tag
    # the three directions one after another are the transform
    dst = np.zeros_like(v)
    for ap in (10, 11, 12):
        dwt.cdf97_3f_op_wrapper_s(dwt.volume_of(src), dwt.volume_of(dst), ap)
    assert np.array_equal(bits(dst), bits(full))
    assert np.array_equal(bits(src), bits(v))


def test_volume_perftest_protocol(dwt, tmp_path):
    """volume_perftest_fwd97op_s(256, ...) -- the reference's 3-D perf test (src/volume-dwt.c:2810) --
    returns 0 errors through ctypes and from a C program written against include/volume-dwt.h."""
    import os
    import subprocess

    err, secs = dwt.volume_perftest_fwd97op_s(256, 1, 0, 2)
    assert err == 0 and 0 < secs < 1e-6
    err, secs_dev = dwt.volume_perftest_fwd97op_s(256, 0, 0, 3, device=True)
    assert err == 0 and 0 < secs_dev < secs
    err, _ = dwt.volume_perftest_fwd97op_s(40, 2, 7, 1)  # VOL_HORIZ_VERT4X4X4
    assert err == 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "volume_perftest"
    libdir = os.path.join(root, "libdwt_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "volume_perftest.c"), "-o", str(exe),
                           "-L", libdir, "-l:libdwt_hip.so", "-Wl,-rpath," + libdir, "-lm"])
    out = subprocess.run([str(exe), "256"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert "volume perftest: success (0 errors)" in out.stderr
    assert "host volumes" in out.stderr and "device volumes" in out.stderr


def test_one_pass_levels_randomised(dwt, oracle):
    """Seeded soak of the one-pass 3-D levels (vol_fused = 2 forces them wherever they can run): random
    sizes from 2 x 2 x 8 up (tiles overhanging by any amount, one or several tile rows / columns),
    1-3 levels, marches of 4 .. 20 slice pairs; in place forward and inverse (k_vol_level_ip over its
    shell, levels >= 1 of the inverse out of place into the lattice above) and out of place forward
    (k_vol_fwd_fused): the oracle's bits every time."""
    rng = np.random.default_rng(20261004)
    dwt.set_option("vol_fused", 2)
    done = 0
    try:
        for case in range(140):
            nz = int(rng.integers(8, 72))
            ny = int(rng.choice([2, 3, 31, 32, 33, 64, 65, 100])) if rng.random() < 0.5 else int(rng.integers(2, 140))
            nx = int(rng.choice([2, 5, 255, 256, 257, 260, 511, 512, 513, 520])) if rng.random() < 0.5 else int(rng.integers(2, 600))
            levels = int(rng.integers(1, 4))
            while levels > 1 and min(-(-n // (1 << (levels - 1))) for n in (nx, ny, nz)) < 2:
                levels -= 1
            tp = int(rng.choice([0, 4, 5, 8, 13, 20]))
            waves = int(rng.choice([4, 8]))  # tiles of 32 rows (two workgroups per CU) or of 64 rows (one)
            dwt.set_option("vol_tile_pairs", tp)
            dwt.set_option("vol_ip_waves", waves)
            vol = rng.random((nz, ny, nx), dtype=np.float32)
            want = oracle_multilevel(oracle, vol.copy(), levels, False)
            what = f"case {case}: {nz}x{ny}x{nx} levels {levels} march {tp} waves {waves}"
            d = DevVol(dwt, vol)
            d.run(0, levels)
            assert np.array_equal(bits(d.get()), bits(want)), what + " (in place, forward)"
            d.run(1, levels)
            assert np.array_equal(bits(d.get()), bits(oracle_multilevel(oracle, want.copy(), levels, True))), what + " (in place, inverse)"
            if case % 3 == 0:
                s, o = DevVol(dwt, vol), DevVol(dwt, np.zeros_like(vol))
                dwt.transform3d_op(s.ptr, o.ptr, nx * 4, nx * ny * 4, nx, ny, nz, levels)
                assert np.array_equal(bits(o.get()), bits(want)), what + " (out of place)"
                s.free()
                o.free()
            d.free()
            done += 1
    finally:
        dwt.set_option("vol_fused", 1)
        dwt.set_option("vol_tile_pairs", 0)
        dwt.set_option("vol_ip_waves", 0)
    assert done == 140


def test_volume_entry_errors(dwt):
    """The field-level entries refuse what they cannot do -- non-zero return and a message, nothing
    transformed -- as the typed wrappers then log and abort (dwt_util_error semantics)."""
    L = dwt.lib
    v = np.random.default_rng(3).random((9, 10, 12), dtype=np.float32)
    keep = v.copy()
    sy, sz = 12 * 4, 12 * 10 * 4
    assert L.dwt_hip_volume_fwd_op(v.ctypes.data, sy, sz, v.ctypes.data, sy, sz, 12, 10, 9, 7) != 0  # out of place only
    assert "out of place" in dwt.last_error()
    out = np.zeros_like(v)
    assert L.dwt_hip_volume_fwd_op(v.ctypes.data, sy, sz, out.ctypes.data, sy, sz, 12, 10, 9, 3) != 0  # dirs
    assert L.dwt_hip_volume_fwd_op(v.ctypes.data, sy - 4, sz, out.ctypes.data, sy, sz, 12, 10, 9, 7) != 0  # rows overlap
    assert L.dwt_hip_volume_ip(0, None, sy, sz, 12, 10, 9) != 0
    d = DevVol(dwt, v)
    assert L.dwt_hip_volume_ip(0, d.ptr, sy + 2, (sy + 2) * 10, 12, 10, 9) != 0  # device strides must be whole samples
    assert "multiples of 4" in dwt.last_error()
    d.free()
    assert np.array_equal(v, keep) and not out.any()
