"""GPU parity tests: every call goes through the C-ABI of libdwt_hip.so and is compared
with the committed golden vectors and with the oracle on the same seeded inputs.
Bar: bit-exact (int 5/3 exact; float kernels keep the reference's unfused mul/add
order, so float coefficients are bit-identical too -- far inside the 1e-5 relative
tolerance the north star allows, which is also asserted explicitly)."""
import numpy as np
import pytest

from conftest import bits, full_range_ints, golden_cases, multichannel_cases

pytestmark = pytest.mark.gpu

NAMES = {"cdf97_s": ("cdf97_2f_s", "cdf97_2i_s", np.float32),
         "cdf53_i": ("cdf53_2f_i", "cdf53_2i_i", np.int32),
         "cdf53_s": ("cdf53_2f_s", "cdf53_2i_s", np.float32),
         "cdf97_d": ("cdf97_2f_d", "cdf97_2i_d", np.float64),
         "cdf53_d": ("cdf53_2f_d", "cdf53_2i_d", np.float64),
         "cdf97_i": ("cdf97_2f_i", "cdf97_2i_i", np.int32)}


@pytest.fixture(scope="module")
def dwt():
    import libdwt_amd as d

    d.dwt_util_init()
    yield d
    for k, v in (("generic", 0), ("cpt", 0), ("tile_pairs", 0), ("waves", 4), ("xcd_swizzle", 1), ("ring", 0),
                 ("nt", 7), ("ring_inv", 8), ("fused_d", 1)):
        d.set_option(k, v)
    d.dwt_util_finish()


def rand_img(rng, h, w, dt):
    if dt == np.float32:
        return rng.random((h, w), dtype=np.float32) * 2 - 1
    if dt == np.float64:
        return rng.random((h, w)) * 2 - 1
    return rng.integers(-32768, 32768, size=(h, w), dtype=np.int32)


def rel_err(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(1e-30, np.abs(b).max()))


# ---- golden vectors through the host-pointer (drop-in) entry -------------------------
def _golden_params():
    for w in NAMES:
        for case in golden_cases(w):
            yield pytest.param(w, case, id=f"{w}-{case[0]['name']}")


@pytest.mark.parametrize("accel", [0, 1])
@pytest.mark.parametrize("wname,case", list(_golden_params()))
def test_golden_host_entry(dwt, wname, case, accel):
    meta, src, fwd, inv = case
    dwt.dwt_util_set_accel(accel)
    try:
        if meta.get("s2"):
            h, w = src.shape
            dst = np.full_like(src, meta["dst_fill"])
            j = dwt.dwt_cdf97_2f_s2(src.copy(), dst, dst.strides[0], 4, w, h, w, h, meta["j_in"])
            assert j == meta["j_out"]
            assert np.array_equal(bits(dst), bits(fwd))
            rec = np.full_like(src, meta["rec_fill"])
            dwt.dwt_cdf97_2i_s2(dst, rec, rec.strides[0], 4, w, h, w, h, j)
            assert np.array_equal(bits(rec), bits(inv))
            return
        buf = src.copy()
        es = buf.dtype.itemsize
        (sox, soy), (six, siy) = meta["size_o"], meta["size_i"]
        j = dwt.FORWARD[wname](buf, buf.strides[0], es, sox, soy, six, siy, meta["j_in"],
                               meta["decompose_one"], meta["zero_padding"])
        assert j == meta["j_out"]
        assert np.array_equal(bits(buf), bits(fwd)), "forward differs from the reference's coefficients"
        dwt.INVERSE[wname](buf, buf.strides[0], es, sox, soy, six, siy, j, meta["decompose_one"], meta["zero_padding"])
        assert np.array_equal(bits(buf), bits(inv)), "inverse differs from the reference's output"
    finally:
        dwt.dwt_util_set_accel(0)


@pytest.mark.parametrize("case", multichannel_cases(), ids=lambda c: c[0]["name"])
def test_golden_multichannel_host_entry(dwt, case):
    """The calling convention of the reference's OpenCV wrapper (src/cvdwt.cpp:98-135): one channel
    of an interleaved multi-channel host image, ptr = data + elemSize1*channel, stride_x = step,
    stride_y = elemSize = channels*sizeof(T), inner size and flags as given.  Bit-identical to what
    the reference produced, other channels and pitch padding untouched."""
    meta, src, fwd, inv = case
    wname = meta["wavelet"]
    buf = src.copy()
    es = buf.dtype.itemsize
    (sox, soy), (six, siy) = meta["size_o"], meta["size_i"]
    ptr = buf.ctypes.data + es * meta["channel"]
    j = dwt.FORWARD[wname](ptr, buf.strides[0], buf.strides[1], sox, soy, six, siy, meta["j_in"],
                           meta["decompose_one"], meta["zero_padding"])
    assert j == meta["j_out"]
    assert np.array_equal(bits(buf), bits(fwd)), "forward differs from the reference's coefficients"
    dwt.INVERSE[wname](ptr, buf.strides[0], buf.strides[1], sox, soy, six, siy, j, meta["decompose_one"], meta["zero_padding"])
    assert np.array_equal(bits(buf), bits(inv)), "inverse differs from the reference's output"


def test_multichannel_all_channels_like_cv_dwt_transform(dwt, oracle):
    """dwt::transform of the wrapper (src/cvdwt.cpp:303-350) loops over the channels of one Mat:
    a 3-channel 512x384 float image, every channel transformed in place through the host entry,
    equals the oracle channel by channel; then the inverse restores the image."""
    rng = np.random.default_rng(8)
    h, w, c = 384, 512, 3
    img = rng.random((h, w, c), dtype=np.float32)
    want = img.copy()
    got = img.copy()
    for ch in range(c):
        jw = oracle.call_channel("cdf97_2f_s", want, ch, 4)
        jg = dwt.dwt_cdf97_2f_s(got.ctypes.data + 4 * ch, got.strides[0], got.strides[1], w, h, w, h, 4)
        assert jg == jw == 4
    assert np.array_equal(bits(got), bits(want))
    for ch in range(c):
        dwt.dwt_cdf97_2i_s(got.ctypes.data + 4 * ch, got.strides[0], got.strides[1], w, h, w, h, 4)
    assert np.abs(got - img).max() < 1e-5


# ---- device-resident images vs the oracle ---------------------------------------------
SHAPES = [(512, 512), (1000, 1000), (300, 513), (64, 2048), (2050, 130), (768, 1024), (5, 7), (2, 2), (1536, 2048)]


@pytest.mark.parametrize("wname", list(NAMES))
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("inplace", [True, False], ids=["inplace", "s2"])
def test_device_resident_vs_oracle(dwt, oracle, wname, shape, inplace):
    ff, fi, dt = NAMES[wname]
    h, w = shape
    rng = np.random.default_rng(h * 7919 + w)
    img = rand_img(rng, h, w, dt)
    want = img.copy()
    jw = oracle.fwd(ff, want, -1)
    wid = dwt.WAVELET_ID[wname]
    es = img.dtype.itemsize
    # (double precision: the reference has in-place entries only; the device-level ABI takes src != dst
    # for every wavelet, and the out-of-place call is the one that avoids the in-place detour)
    a = dwt.DeviceImage(h, w, itemsize=es).upload(img)
    b = a if inplace else dwt.DeviceImage(h, w, itemsize=es).upload(np.zeros_like(img))
    j = dwt._fwd(wid, a.ptr, b.ptr, a.stride_x, es, w, h, w, h, -1, 0, 0, "fwd")
    got = b.download(dt)
    assert j == jw
    assert np.array_equal(bits(got), bits(want))
    if dt == np.float32:
        assert rel_err(got, want) <= 1e-5  # the north star's stated tolerance
    if not inplace:
        assert np.array_equal(a.download(dt), img), "_s2 must leave the source untouched"
    # inverse (b -> a for s2, in place otherwise)
    dwt._inv(wid, b.ptr, a.ptr, a.stride_x, es, w, h, w, h, j, 0, 0, "inv")
    rec = a.download(dt)
    oracle.inv(fi, want, jw)
    assert np.array_equal(bits(rec), bits(want))
    if dt == np.int32:
        assert np.array_equal(rec, img), "int 5/3 must reconstruct exactly"
    else:
        assert np.abs(rec - img).max() < 1e-4
    a.free()
    if b is not a:
        b.free()


def test_padded_device_pitch(dwt, oracle):
    """Device image whose row pitch is wider than the image (and not 16-byte aligned)."""
    h, w, pe = 200, 333, 341
    rng = np.random.default_rng(3)
    buf = rng.random((h, pe), dtype=np.float32)
    want = buf.copy()
    jw = oracle.fwd("cdf97_2f_s", want[:, :w], 4)
    d = dwt.DeviceImage(h, w, pitch_bytes=pe * 4).upload(buf)
    j = dwt.dwt_cdf97_2f_s(d.ptr, pe * 4, 4, w, h, w, h, 4)
    got = d.download(np.float32)
    assert j == jw and np.array_equal(bits(got), bits(want))
    d.free()


# ---- every tile geometry gives the same bits --------------------------------------------
@pytest.mark.parametrize("wname", ["cdf97_s", "cdf53_i", "cdf97_i"])
def test_tile_variants_agree(dwt, oracle, wname):
    ff, fi, dt = NAMES[wname]
    h, w = 1100, 1300
    rng = np.random.default_rng(17)
    img = rand_img(rng, h, w, dt)
    want = img.copy()
    jw = oracle.fwd(ff, want, 4)
    rec_want = want.copy()
    oracle.inv(fi, rec_want, jw)
    wid = dwt.WAVELET_ID[wname]
    try:
        for cpt in (4, 8):
            for tp in (8, 24, 128):
                for waves in (1, 4):
                    for swz in (0, 1):
                        dwt.set_option("cpt", cpt)
                        dwt.set_option("tile_pairs", tp)
                        dwt.set_option("waves", waves)
                        dwt.set_option("xcd_swizzle", swz)
                        a = dwt.DeviceImage(h, w).upload(img)
                        b = dwt.DeviceImage(h, w).upload(np.zeros_like(img))
                        j = dwt._fwd(wid, a.ptr, b.ptr, a.stride_x, 4, w, h, w, h, 4, 0, 0, "fwd")
                        got = b.download(dt)
                        assert j == jw and np.array_equal(bits(got), bits(want)), (cpt, tp, waves, swz)
                        dwt._inv(wid, b.ptr, a.ptr, a.stride_x, 4, w, h, w, h, j, 0, 0, "inv")
                        assert np.array_equal(bits(a.download(dt)), bits(rec_want)), (cpt, tp, waves, swz)
                        a.free()
                        b.free()
    finally:
        for k, v in (("cpt", 0), ("tile_pairs", 0), ("waves", 4), ("xcd_swizzle", 1)):
            dwt.set_option(k, v)


def test_ring_layout_and_cache_policy_variants_agree(dwt, oracle):
    """ring depth (with it the layout of a workgroup's waves), cache policy and the wavefront-shift variant of
    the neighbour taps change scheduling only, never the bits."""
    h, w = 700, 2100
    rng = np.random.default_rng(23)
    img = rand_img(rng, h, w, np.float32)
    want = img.copy()
    jw = oracle.fwd("cdf97_2f_s", want, 3)
    rec_want = want.copy()
    oracle.inv("cdf97_2i_s", rec_want, jw)
    try:
        for ring in (8, 16):
            for nt in (3, 7, 15):
                for cpt in (4, 8):
                    for k, v in (("ring", ring), ("ring_inv", ring), ("nt", nt), ("cpt", cpt), ("tile_pairs", 16)):
                        dwt.set_option(k, v)
                    a = dwt.DeviceImage(h, w).upload(img)
                    b = dwt.DeviceImage(h, w).upload(np.zeros_like(img))
                    j = dwt.dwt_cdf97_2f_s2(a.ptr, b.ptr, a.stride_x, 4, w, h, w, h, 3)
                    assert j == jw and np.array_equal(bits(b.download(np.float32)), bits(want)), (ring, nt, cpt)
                    dwt.dwt_cdf97_2i_s2(b.ptr, a.ptr, a.stride_x, 4, w, h, w, h, j)
                    assert np.array_equal(bits(a.download(np.float32)), bits(rec_want)), (ring, nt, cpt)
                    a.free()
                    b.free()
    finally:
        for k, v in (("ring", 0), ("ring_inv", 8), ("nt", 7), ("cpt", 0), ("tile_pairs", 0)):
            dwt.set_option(k, v)


def test_config1_simple_example_flow(dwt, oracle):
    """configs[0]: 512x512 float, libdwt pattern, prime pitch 2053 B, as examples/simple."""
    x = y = 512
    stride_y = 4
    stride_x = dwt.lib.dwt_util_get_opt_stride(stride_y * x)
    assert stride_x == 2053
    raw1 = np.zeros(stride_x * y + 16, np.uint8)
    dwt.lib.dwt_util_test_image_fill_s(raw1.ctypes.data, stride_x, stride_y, x, y, 0)
    raw2 = raw1.copy()
    ref = raw1.copy()
    for jreq in (1, -1):
        raw1[:] = raw2
        ref[:] = raw2
        j = dwt.dwt_cdf97_2f_s(raw1.ctypes.data, stride_x, stride_y, x, y, x, y, jreq)
        import ctypes as C
        jr = C.c_int(jreq)
        oracle.lib.oracle_cdf97_2f_s(ref.ctypes.data, stride_x, stride_y, x, y, x, y, C.byref(jr), 0, 0)
        assert j == jr.value == (1 if jreq == 1 else 9)
        assert np.array_equal(raw1, ref)
        dwt.dwt_cdf97_2i_s(raw1.ctypes.data, stride_x, stride_y, x, y, x, y, j)
        assert dwt.lib.dwt_util_compare_s(raw1.ctypes.data, raw2.ctypes.data, stride_x, stride_y, x, y) == 0
        # round-trip PSNR (peak 1.0)
        a = np.array([np.frombuffer(raw1[r * stride_x:r * stride_x + 4 * x].tobytes(), np.float32) for r in range(y)])
        b = np.array([np.frombuffer(raw2[r * stride_x:r * stride_x + 4 * x].tobytes(), np.float32) for r in range(y)])
        mse = float(np.mean((a.astype(np.float64) - b) ** 2))
        psnr = 10 * np.log10(1.0 / max(mse, 1e-30))
        assert psnr > 100.0, psnr


def test_config2_8192_full_size(dwt, oracle):
    """configs[1]: 8192x8192 float, 5 levels, device resident, full compare with the oracle."""
    n = 8192
    rng = np.random.default_rng(1234)
    img = rng.random((n, n), dtype=np.float32)
    a = dwt.DeviceImage(n, n).upload(img)
    b = dwt.DeviceImage(n, n)
    j = dwt.dwt_cdf97_2f_s2(a.ptr, b.ptr, n * 4, 4, n, n, n, n, 5)
    got = b.download(np.float32)
    want = img.copy()
    assert oracle.fwd("cdf97_2f_s", want, 5) == j == 5
    assert np.array_equal(bits(got), bits(want))
    # in-place entry on the same data
    j = dwt.dwt_cdf97_2f_s(a.ptr, n * 4, 4, n, n, n, n, 5)
    assert np.array_equal(bits(a.download(np.float32)), bits(want))
    # size-independent properties: round trip and energy of the LL band (DC gain 2^J)
    dwt.dwt_cdf97_2i_s(a.ptr, n * 4, 4, n, n, n, n, 5)
    rec = a.download(np.float32)
    assert np.abs(rec - img).max() < 1e-4
    ll = got[: n >> 5, : n >> 5]
    assert abs(ll.mean() / (img.mean() * 32) - 1) < 1e-3
    a.free()
    b.free()


def test_config3_int53_4096(dwt, oracle):
    """configs[2]: 4096x4096 int32, 3 levels: bit-exact forward, exact reconstruction."""
    n = 4096
    rng = np.random.default_rng(7)
    img = rng.integers(-32768, 32768, size=(n, n), dtype=np.int32)
    a = dwt.DeviceImage(n, n).upload(img)
    j = dwt.dwt_cdf53_2f_i(a.ptr, n * 4, 4, n, n, n, n, 3)
    want = img.copy()
    assert oracle.fwd("cdf53_2f_i", want, 3) == j == 3
    assert np.array_equal(a.download(np.int32), want)
    dwt.dwt_cdf53_2i_i(a.ptr, n * 4, 4, n, n, n, n, 3)
    assert np.array_equal(a.download(np.int32), img)
    # libdwt's own int pattern through the host entry (examples/simple-int)
    pat = np.zeros((512, 512), np.int32)
    dwt.lib.dwt_util_test_image_fill_i(pat.ctypes.data, 2048, 4, 512, 512, 0)
    keep = pat.copy()
    want = pat.copy()
    j = dwt.dwt_cdf53_2f_i(pat, 2048, 4, 512, 512, 512, 512, -1)
    assert oracle.fwd("cdf53_2f_i", want, -1) == j
    assert np.array_equal(pat, want)
    dwt.dwt_cdf53_2i_i(pat, 2048, 4, 512, 512, 512, 512, j)
    assert dwt.lib.dwt_util_compare_i(pat.ctypes.data, keep.ctypes.data, 2048, 4, 512, 512) == 0
    a.free()


def test_config4_batch(dwt, oracle):
    """configs[3] at reduced count: a batch of independent images in one launch per level."""
    n, nb = 1024, 6
    rng = np.random.default_rng(99)
    imgs = rng.random((nb, n, n), dtype=np.float32)
    src = dwt.lib.dwt_hip_malloc(imgs.nbytes)
    dst = dwt.lib.dwt_hip_malloc(imgs.nbytes)
    assert src and dst
    assert dwt.lib.dwt_hip_memcpy_h2d(src, imgs.ctypes.data, imgs.nbytes) == 0
    j = dwt.transform2d_batch("cdf97_s", 0, src, dst, n * n * 4, nb, n * 4, n, n, 5)
    out = np.empty_like(imgs)
    assert dwt.lib.dwt_hip_memcpy_d2h(out.ctypes.data, dst, imgs.nbytes) == 0
    assert j == 5
    for k in range(nb):
        want = imgs[k].copy()
        oracle.fwd("cdf97_2f_s", want, 5)
        assert np.array_equal(bits(out[k]), bits(want)), k
    # inverse of the batch back into src
    dwt.transform2d_batch("cdf97_s", 1, dst, src, n * n * 4, nb, n * 4, n, n, 5)
    assert dwt.lib.dwt_hip_memcpy_d2h(out.ctypes.data, src, imgs.nbytes) == 0
    assert np.abs(out - imgs).max() < 1e-4
    dwt.lib.dwt_hip_free(src)
    dwt.lib.dwt_hip_free(dst)


def test_host_channel_calls_from_several_threads(dwt, oracle):
    """Host-pointer calls on channels of interleaved images from three threads at once: the strided
    repacking runs on the library's row pool, whose jobs take turns; every result is the oracle's."""
    import threading

    errors = []

    def work(i):
        try:
            dwt.set_device(0)
            rng = np.random.default_rng(500 + i)
            h, w, c = 700 + 64 * i, 900, 3
            for rep in range(3):
                img = rng.random((h, w, c), dtype=np.float32)
                want = img.copy()
                ch = (i + rep) % c
                j = oracle.call_channel("cdf97_2f_s", want, ch, 3)
                got = img.copy()
                assert dwt.dwt_cdf97_2f_s(got.ctypes.data + 4 * ch, got.strides[0], got.strides[1], w, h, w, h, 3) == j
                if not np.array_equal(bits(got), bits(want)):
                    errors.append((i, rep))
            dwt.dwt_util_finish()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_double_precision_batch(dwt, oracle):
    """The batched entry takes the double-precision wavelets too (fused double sweeps)."""
    n, nb = 768, 3
    rng = np.random.default_rng(4)
    imgs = rng.random((nb, n, n)) * 2 - 1
    src = dwt.lib.dwt_hip_malloc(imgs.nbytes)
    dst = dwt.lib.dwt_hip_malloc(imgs.nbytes)
    try:
        assert dwt.lib.dwt_hip_memcpy_h2d(src, imgs.ctypes.data, imgs.nbytes) == 0
        for wname, ff in (("cdf97_d", "cdf97_2f_d"), ("cdf53_d", "cdf53_2f_d")):
            j = dwt.transform2d_batch(wname, 0, src, dst, n * n * 8, nb, n * 8, n, n, 4)
            out = np.empty_like(imgs)
            assert dwt.lib.dwt_hip_memcpy_d2h(out.ctypes.data, dst, imgs.nbytes) == 0
            assert j == 4
            for k in range(nb):
                want = imgs[k].copy()
                oracle.fwd(ff, want, 4)
                assert np.array_equal(bits(out[k]), bits(want)), (wname, k)
    finally:
        dwt.lib.dwt_hip_free(src)
        dwt.lib.dwt_hip_free(dst)


BATCH_D_SCRIPT = r"""
import sys, numpy as np
import torch                      # first: this process then shares torch's HIP runtime
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from libdwt_amd import batch as B
from oraclelib import Oracle
n, nb = 384, 3
imgs = np.random.default_rng(4).random((nb, n, n)) * 2 - 1
got = B.transform_sharded(torch.from_numpy(imgs).cuda(), nb, (n, n), "cdf97_d", 4)
torch.cuda.synchronize()
orc = Oracle()
for k in range(nb):
    want = imgs[k].copy()
    orc.fwd("cdf97_2f_d", want, 4)
    assert np.array_equal(got[k].cpu().numpy().view(np.uint64), want.view(np.uint64)), k
print("batch_d OK")
"""


def test_python_batch_split_keeps_double_precision_strides():
    """libdwt_amd/batch.py (here one rank) with a double-precision wavelet: 8-byte element strides all the way
    (round 3 passed `h*w*4`).  Own process: torch before the library, so that both use one HIP runtime."""
    import os
    import subprocess
    import sys

    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + BATCH_D_SCRIPT], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "batch_d OK" in out.stdout, out.stderr[-2000:]


def test_config4_per_gpu_shape_32x4096(dwt, oracle):
    """configs[3] at its per-GPU scale: 32 images of 4096x4096, 5 levels, ONE
    batched call (one launch per level for the shard of a GPU); image k seeded 1234+k; images 0
    and 31 (first and last of rank 0's block b*G//B of the 256) bit-compared with the oracle,
    every other image through size-independent checks."""
    n, nb, J = 4096, 32, 5
    imgs = np.empty((nb, n, n), np.float32)
    for k in range(nb):
        imgs[k] = np.random.default_rng(1234 + k).random((n, n), dtype=np.float32)
    src = dwt.lib.dwt_hip_malloc(imgs.nbytes)
    dst = dwt.lib.dwt_hip_malloc(imgs.nbytes)
    one = dwt.DeviceImage(n, n)
    assert src and dst
    try:
        assert dwt.lib.dwt_hip_memcpy_h2d(src, imgs.ctypes.data, imgs.nbytes) == 0
        j = dwt.transform2d_batch("cdf97_s", 0, src, dst, n * n * 4, nb, n * 4, n, n, J)
        assert j == J
        out = np.empty_like(imgs)
        assert dwt.lib.dwt_hip_memcpy_d2h(out.ctypes.data, dst, imgs.nbytes) == 0
        for k in (0, 31):
            want = imgs[k].copy()
            assert oracle.fwd("cdf97_2f_s", want, J) == J
            assert np.array_equal(bits(out[k]), bits(want)), f"image {k} differs from the oracle"
        # every image: the LL band carries the DC gain 2^J ...
        ll = out[:, : n >> J, : n >> J].mean(axis=(1, 2), dtype=np.float64)
        mean = imgs.mean(axis=(1, 2), dtype=np.float64)
        assert np.all(np.abs(ll / (mean * (1 << J)) - 1) < 1e-3)
        # ... the batched call equals the single-image libdwt.h entry bit for bit ...
        dwt.dwt_cdf97_2f_s2(src + 17 * n * n * 4, one.ptr, n * 4, 4, n, n, n, n, J)
        assert np.array_equal(bits(one.download(np.float32)), bits(out[17]))
        # ... and the inverse of the whole shard restores it
        dwt.transform2d_batch("cdf97_s", 1, dst, src, n * n * 4, nb, n * 4, n, n, J)
        assert dwt.lib.dwt_hip_memcpy_d2h(out.ctypes.data, src, imgs.nbytes) == 0
        assert np.abs(out - imgs).max() < 1e-4
    finally:
        dwt.lib.dwt_hip_free(src)
        dwt.lib.dwt_hip_free(dst)
        one.free()


def test_config4_images_of_ranks_1_and_7(dwt, oracle):
    """SURVEY.md s8d C4 names the images 0, 31, 32 and 255 of the 256: 0 and 31 are rank 0's first and last (test
    above); 32 is the FIRST image of rank 1's block and 255 the LAST of rank 7's, found through
    dwt_hip_shard_bounds(256, 8, rank) as a rank of the sharded run finds them.  Each rank's block head / tail
    (two images around them) goes through one batched call; 32 and 255 are bit-compared with the oracle."""
    n, J, total, world = 4096, 5, 256, 8
    a1, c1 = dwt.shard_bounds(total, world, 1)
    a7, c7 = dwt.shard_bounds(total, world, 7)
    assert (a1, c1) == (32, 32) and (a7, c7) == (224, 32)
    for first, pick in ((a1, 0), (a7 + c7 - 2, 1)):  # images [32, 33] and [254, 255]
        imgs = np.empty((2, n, n), np.float32)
        for i in range(2):
            imgs[i] = np.random.default_rng(1234 + first + i).random((n, n), dtype=np.float32)
        src = dwt.lib.dwt_hip_malloc(imgs.nbytes)
        dst = dwt.lib.dwt_hip_malloc(imgs.nbytes)
        try:
            assert dwt.lib.dwt_hip_memcpy_h2d(src, imgs.ctypes.data, imgs.nbytes) == 0
            assert dwt.transform2d_batch("cdf97_s", 0, src, dst, n * n * 4, 2, n * 4, n, n, J) == J
            out = np.empty_like(imgs)
            assert dwt.lib.dwt_hip_memcpy_d2h(out.ctypes.data, dst, imgs.nbytes) == 0
            want = imgs[pick].copy()
            assert oracle.fwd("cdf97_2f_s", want, J) == J
            assert np.array_equal(bits(out[pick]), bits(want)), f"image {first + pick} differs from the oracle"
        finally:
            dwt.lib.dwt_hip_free(src)
            dwt.lib.dwt_hip_free(dst)


STREAMS_SCRIPT = r"""
import sys, numpy as np
import torch                      # first: this process then shares torch's HIP runtime
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import libdwt_amd as dwt
from oraclelib import Oracle
dwt.dwt_util_init()
orc = Oracle()
n, J, reps = 2048, 4, 6
rng = np.random.default_rng(321)
imgs = [rng.random((n, n), dtype=np.float32) for _ in range(2 * reps)]
wants = []
for a in imgs:
    w_ = a.copy(); orc.fwd("cdf97_2f_s", w_, J); wants.append(w_)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
dev = [torch.from_numpy(a).to("cuda:0") for a in imgs]
torch.cuda.synchronize()
for i, t in enumerate(dev):
    with torch.cuda.stream(streams[i & 1]):
        dwt.use_torch_stream()
        dwt.dwt_cdf97_2f_s(t, n * 4, 4, n, n, n, n, J)
torch.cuda.synchronize()
for i, t in enumerate(dev):
    assert np.array_equal(t.cpu().numpy().view(np.uint32), wants[i].view(np.uint32)), f"forward call {i} (stream {i & 1})"
for i, t in enumerate(dev):   # and back, alternating the other way round
    with torch.cuda.stream(streams[(i + 1) & 1]):
        dwt.use_torch_stream()
        dwt.dwt_cdf97_2i_s(t, n * 4, 4, n, n, n, n, J)
torch.cuda.synchronize()
for i, t in enumerate(dev):
    assert np.abs(t.cpu().numpy() - imgs[i]).max() < 1e-5, f"inverse call {i}"
# host-pointer calls (pinned staging, host_a / host_b) between two streams as well
h = [rng.random((700, 900), dtype=np.float32) for _ in range(4)]
for i, a in enumerate(h):
    w_ = a.copy(); orc.fwd("cdf97_2f_s", w_, 3)
    with torch.cuda.stream(streams[i & 1]):
        dwt.use_torch_stream()
        dwt.dwt_cdf97_2f_s(a, a.strides[0], 4, 900, 700, 900, 700, 3)
    assert np.array_equal(a.view(np.uint32), w_.view(np.uint32)), i
print("streams OK")
"""


def test_two_streams_alternating_on_one_thread():
    """dwt_hip_set_stream orders a newly set stream behind what the context queued on the old one (the LL scratch,
    the staging image and the pinned buffer are shared by the context's streams).  Two torch streams alternate call
    by call on one thread -- in-place entries (staging + scratch), different images per stream, no synchronisation
    in between -- and every result is the oracle's.  Own process: torch before the library (one HIP runtime)."""
    import os
    import subprocess
    import sys

    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + STREAMS_SCRIPT], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "streams OK" in out.stdout, out.stderr[-2000:]


def test_threads_have_their_own_context(dwt, oracle):
    """One context per host thread: four threads transform different images at the same time (device
    pointers and host pointers, different sizes, so their workspaces differ) and every result is
    the oracle's; dwt_hip_set_device binds a thread, an out-of-range device is refused."""
    import threading

    assert dwt.get_device() == 0
    shapes = [(1024, 1024), (768, 1280), (2048, 512), (640, 640)]
    results, errors = {}, []

    def work(i):
        try:
            dwt.set_device(0)
            h, w = shapes[i]
            rng = np.random.default_rng(100 + i)
            for rep in range(6):
                img = rng.random((h, w), dtype=np.float32)
                want = img.copy()
                j = oracle.fwd("cdf97_2f_s", want, 4)
                if rep % 2:
                    got = img.copy()
                    assert dwt.dwt_cdf97_2f_s(got, got.strides[0], 4, w, h, w, h, 4) == j
                else:
                    d = dwt.DeviceImage(h, w).upload(img)
                    assert dwt.dwt_cdf97_2f_s(d.ptr, d.stride_x, 4, w, h, w, h, 4) == j
                    got = d.download(np.float32)
                    d.free()
                if not np.array_equal(bits(got), bits(want)):
                    errors.append((i, rep))
            results[i] = True
            dwt.dwt_util_finish()  # this thread's workspace
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert len(results) == 4
    with pytest.raises(dwt.DwtError):
        dwt.set_device(dwt.device_count() + 3)
    dwt.set_device(0)


def test_linearity_and_constant(dwt):
    n = 2048
    rng = np.random.default_rng(5)
    x = rng.random((n, n), dtype=np.float32)
    y = rng.random((n, n), dtype=np.float32)
    outs = []
    for img in (x, y, (x + y).astype(np.float32)):
        d = dwt.DeviceImage(n, n).upload(img)
        dwt.dwt_cdf97_2f_s(d.ptr, n * 4, 4, n, n, n, n, 5)
        outs.append(d.download(np.float32))
        d.free()
    assert np.abs(outs[0] + outs[1] - outs[2]).max() < 2e-3
    c = np.full((n, n), 3.0, np.float32)
    d = dwt.DeviceImage(n, n).upload(c)
    dwt.dwt_cdf97_2f_s(d.ptr, n * 4, 4, n, n, n, n, 5)
    t = d.download(np.float32)
    d.free()
    ll = n >> 5
    assert np.allclose(t[:ll, :ll], 3.0 * 32, rtol=1e-5)
    mask = np.ones_like(t, bool)
    mask[:ll, :ll] = False
    assert np.abs(t[mask]).max() < 1e-4


def test_errors_are_reported(dwt):
    img = np.zeros((8, 8), np.float32)
    with pytest.raises(dwt.DwtError):
        dwt._fwd(7, img, img, 32, 4, 8, 8, 8, 8, -1, 0, 0, "bad wavelet")
    d = dwt.DeviceImage(8, 8)
    with pytest.raises(dwt.DwtError):  # mixing host and device pointers
        dwt._fwd(0, img, d.ptr, 32, 4, 8, 8, 8, 8, -1, 0, 0, "mixed")
    d.free()


def test_c_example_program(dwt, tmp_path):
    """examples/roundtrip.c: a C program against include/libdwt.h, host and device images."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "roundtrip"
    libdir = os.path.join(root, "libdwt_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "roundtrip.c"), "-o", str(exe),
                           "-L", libdir, "-l:libdwt_hip.so", "-Wl,-rpath," + libdir, "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "host round trip: success" in out.stderr and "device round trip: success" in out.stderr
    # examples/volume3d.c: the out-of-place 3-D entry and the interleaved-layout 2-D entries from C
    exe = tmp_path / "volume3d"
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "volume3d.c"), "-o", str(exe),
                           "-L", libdir, "-l:libdwt_hip.so", "-Wl,-rpath," + libdir, "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "volume round trip: success" in out.stderr and "interleaved round trip: success" in out.stderr
    # examples/batch_multi.c: the batch split over the node's GPUs from one C process (3 slots: several
    # contexts on the one GPU of the test box), bits equal to the single-GPU batched call
    exe = tmp_path / "batch_multi"
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "batch_multi.c"), "-o", str(exe),
                           "-L", libdir, "-l:libdwt_hip.so", "-Wl,-rpath," + libdir, "-lm"])
    out = subprocess.run([str(exe), "7", "1024", "4", "3"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "bits equal the single-GPU call" in out.stdout, (out.stdout, out.stderr)
    # ... and with every shard resident where it is transformed (dwt_hip_transform2d_batch_multi; 4 slots)
    out = subprocess.run([str(exe), "--resident", "9", "1024", "4", "4"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "shards resident per device" in out.stdout and "bits equal the single-GPU call" in out.stdout, (out.stdout, out.stderr)


def test_harness_helpers(dwt):
    """libdwt's own self-test and perf helpers (src/libdwt.c:24163, 24203, 21391) over the backend."""
    import ctypes as C

    L = dwt.lib
    L.dwt_util_test2_cdf97_2_s.restype = C.c_int
    L.dwt_util_test2_cdf97_2_s2.restype = C.c_int
    # the loop of examples/test/test.c: 256x256, DWT_ARR_SIMPLE (0), opt stride 1, full depth, decompose_one
    for arr in (0, 1, 2):
        assert L.dwt_util_test2_cdf97_2_s(arr, 256, 256, 1, -1, 1) == 0
        assert L.dwt_util_test2_cdf97_2_s2(arr, 256, 256, 1, -1, 1) == 0
        assert L.dwt_util_test2_cdf97_2_s(arr, 200, 120, 1, -1, 1) == 0
    # The reference's own `_s2` self-test FAILS on sparse frames (dwt_cdf97_2i_s2 copies only
    # the inner region, src/libdwt.c:18001-18008, but detail subbands sit at outer-frame
    # offsets); the drop-in reproduces that, bit for bit (checked against libdwt_ref.so: 1, 1, 0)
    assert [L.dwt_util_test2_cdf97_2_s2(arr, 200, 120, 1, -1, 1) for arr in (0, 1, 2)] == [1, 1, 0]
    f, i = C.c_float(0), C.c_float(0)
    L.dwt_util_perf_cdf97_2_s.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_float)] * 2
    L.dwt_util_perf_cdf97_2_s(2048 * 4, 4, 2048, 2048, 2048, 2048, 3, 0, 0, 2, 3, 0, C.byref(f), C.byref(i))
    assert 0 < f.value < 1 and 0 < i.value < 1
    L.dwt_hip_perf_cdf97_2_s.argtypes = [C.c_int] * 12 + [C.POINTER(C.c_float)] * 2
    L.dwt_hip_perf_cdf97_2_s(2048 * 4, 4, 2048, 2048, 2048, 2048, 3, 0, 0, 2, 3, 0, C.byref(f), C.byref(i))
    assert 0 < f.value < 0.01 and 0 < i.value < 0.01
    # subband addressing on a device image: read HH of level 2 back through the address it returns
    n = 64
    img = np.random.default_rng(1).random((n, n), dtype=np.float32)
    d = dwt.DeviceImage(n, n).upload(img)
    dwt.dwt_cdf97_2f_s(d.ptr, n * 4, 4, n, n, n, n, 2)
    full = d.download(np.float32)
    p, sx, sy = C.c_void_p(), C.c_int(), C.c_int()
    L.dwt_util_subband_s.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.dwt_util_subband_s(d.ptr, n * 4, 4, n, n, n, n, 2, 3, C.byref(p), C.byref(sx), C.byref(sy))
    assert (sx.value, sy.value) == (16, 16) and p.value == d.ptr + 16 * n * 4 + 16 * 4
    row = np.empty(16, np.float32)
    assert L.dwt_hip_memcpy_d2h(row.ctypes.data, p.value, 64) == 0
    assert np.array_equal(row, full[16, 16:32])
    d.free()


GRAPH_SCRIPT = r"""
import sys, numpy as np
import torch                      # first: this process then shares torch's HIP runtime
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import libdwt_amd as dwt
from oraclelib import Oracle
dwt.dwt_util_init()
n, nb = 1024, 3
x = torch.rand((nb, n, n), device="cuda"); y = torch.empty_like(x)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    dwt.use_torch_stream()
    dwt.transform2d_batch("cdf97_s", 0, x, y, n * n * 4, nb, n * 4, n, n, 4)   # warm-up: allocates scratch
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        dwt.use_torch_stream()
        dwt.transform2d_batch("cdf97_s", 0, x, y, n * n * 4, nb, n * 4, n, n, 4)
y.zero_(); x.copy_(torch.rand((nb, n, n), device="cuda"))
g.replay(); torch.cuda.synchronize()
want = x[1].cpu().numpy().copy()
Oracle().fwd("cdf97_2f_s", want, 4)
assert np.array_equal(y[1].cpu().numpy().view(np.uint32), want.view(np.uint32)), "graph replay differs"
# the interleaved 9/7 forward forks its exact border strips onto a side stream and joins it again:
# that fork / join is captured with the call
m = 2048
a = torch.rand((m, m), device="cuda"); b = torch.empty_like(a)
with torch.cuda.stream(s):
    dwt.use_torch_stream()
    dwt.transform2d_interleaved("cdf97_s", 0, 0, a, b, m * 4, 4, m, m, None, None, 4)
    s.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        dwt.use_torch_stream()
        dwt.transform2d_interleaved("cdf97_s", 0, 0, a, b, m * 4, 4, m, m, None, None, 4)
b.zero_(); a.copy_(torch.rand((m, m), device="cuda"))
g2.replay(); torch.cuda.synchronize()
want = a.cpu().numpy().copy()
Oracle().fwd("cdf97_2f_inplace_s", want, 4)
assert np.array_equal(b.cpu().numpy().view(np.uint32), want.view(np.uint32)), "graph replay of the interleaved call differs"
print("graph OK")
"""


@pytest.mark.parametrize("w,h,pitch_elems", [(999, 777, 1003), (1000, 500, 1001), (4100, 300, 4101), (515, 64, 517), (8188, 40, 8189)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("wname", ["cdf97_s", "cdf53_i", "cdf97_d"])
def test_device_images_with_unaligned_pitch(dwt, oracle, wname, w, h, pitch_elems):
    """Device images whose rows are only element-aligned (row pitch no multiple of 16 bytes) and odd
    widths: the sweeps address rows as buffers with a per-dword bounds check, so these take the same
    16-byte path as aligned images; bit-exact, the padding between rows untouched."""
    ff, fi, dt = NAMES[wname]
    es = np.dtype(dt).itemsize
    ut = np.uint32 if es == 4 else np.uint64
    fill = ut(0x7B7B7B7B7B7B7B7B & ((1 << (8 * es)) - 1))
    rng = np.random.default_rng(w * 7 + h)
    img = rng.integers(-32768, 32767, (h, w)).astype(np.int32) if dt == np.int32 else rng.random((h, w)).astype(dt)
    J = 4
    want = img.copy()
    oracle.fwd(ff, want, J)
    rec = want.copy()
    oracle.inv(fi, rec, J)
    pad = np.full((h, pitch_elems), fill, ut)
    pad[:, :w] = img.view(ut)
    src = dwt.DeviceImage(h, w, es, pitch_elems * es).upload(pad)
    # in place
    dwt.FORWARD[wname](src.ptr, pitch_elems * es, es, w, h, w, h, J)
    got = src.download(ut)
    assert np.array_equal(got[:, :w], want.view(ut))
    assert np.all(got[:, w:] == fill), "row padding written"
    dwt.INVERSE[wname](src.ptr, pitch_elems * es, es, w, h, w, h, J)
    back = src.download(ut)
    assert np.array_equal(back[:, :w], rec.view(ut))
    assert np.all(back[:, w:] == fill)
    # out of place (the float 9/7 entries have the _s2 form)
    if wname == "cdf97_s":
        dst = dwt.DeviceImage(h, w, es, pitch_elems * es).upload(np.full((h, pitch_elems), fill, ut))
        src.upload(pad)
        dwt.dwt_cdf97_2f_s2(src.ptr, dst.ptr, pitch_elems * es, es, w, h, w, h, J)
        got = dst.download(ut)
        assert np.array_equal(got[:, :w], want.view(ut))
        assert np.all(got[:, w:] == fill)
        dst.free()
    src.free()


def test_hip_graph_capture_replay():
    """After one warm-up call (workspace allocated) a device-resident transform issues only
    kernel launches on the caller's stream, so it can be captured into a HIP graph and
    replayed.  Own process: torch has to be imported before the library so that both use
    one HIP runtime."""
    import os
    import subprocess
    import sys

    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + GRAPH_SCRIPT], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "graph OK" in out.stdout, out.stderr[-2000:]


def test_fma_option_within_tolerance(dwt, oracle):
    """Option "fma" contracts the float 9/7 steps: not bit-identical, but inside the 1e-5
    relative tolerance the north star states for float CDF 9/7."""
    n = 2048
    img = np.random.default_rng(77).random((n, n), dtype=np.float32)
    want = img.copy()
    oracle.fwd("cdf97_2f_s", want, 5)
    try:
        dwt.set_option("fma", 1)
        a = dwt.DeviceImage(n, n).upload(img)
        b = dwt.DeviceImage(n, n)
        dwt.dwt_cdf97_2f_s2(a.ptr, b.ptr, n * 4, 4, n, n, n, n, 5)
        got = b.download(np.float32)
        assert not np.array_equal(bits(got), bits(want))  # it really is a different rounding
        assert rel_err(got, want) <= 1e-5
        dwt.dwt_cdf97_2i_s2(b.ptr, a.ptr, n * 4, 4, n, n, n, n, 5)
        assert np.abs(a.download(np.float32) - img).max() < 1e-4
        a.free()
        b.free()
    finally:
        dwt.set_option("fma", 0)


def test_device_side_conv_show_and_compare(dwt):
    """dwt_hip_conv_show / dwt_hip_compare: the view and the comparison of examples/simple
    (src/libdwt.c:21075, 1593) on images that stay in HBM."""
    import ctypes as C

    L = dwt.lib
    L.dwt_hip_conv_show.argtypes = [C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 4
    L.dwt_hip_conv_show.restype = C.c_int
    L.dwt_hip_compare.argtypes = [C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 4
    L.dwt_hip_compare.restype = C.c_int
    h, w = 300, 500
    rng = np.random.default_rng(2)
    img = (rng.random((h, w), dtype=np.float32) - 0.5) * 8
    a = dwt.DeviceImage(h, w).upload(img)
    b = dwt.DeviceImage(h, w)
    assert L.dwt_hip_conv_show(0, a.ptr, b.ptr, w * 4, 4, w, h) == 0
    host = np.zeros_like(img)
    L.dwt_util_conv_show_s(img.ctypes.data, host.ctypes.data, w * 4, 4, w, h)
    assert np.allclose(b.download(np.float32), host, rtol=2e-6, atol=1e-7)
    ii = rng.integers(-1000, 1000, size=(h, w), dtype=np.int32)
    a.upload(ii)
    assert L.dwt_hip_conv_show(1, a.ptr, b.ptr, w * 4, 4, w, h) == 0
    assert np.array_equal(b.download(np.int32), np.abs(ii))
    # compare
    a.upload(img)
    b.upload(img)
    assert L.dwt_hip_compare(0, a.ptr, b.ptr, w * 4, 4, w, h) == 0
    img2 = img.copy()
    img2[123, 77] += 5e-4
    b.upload(img2)
    assert L.dwt_hip_compare(0, a.ptr, b.ptr, w * 4, 4, w, h) == 0
    img2[123, 77] += 1e-2
    b.upload(img2)
    assert L.dwt_hip_compare(0, a.ptr, b.ptr, w * 4, 4, w, h) == 1
    img2 = img.copy()
    img2[0, 0] = np.nan
    b.upload(img2)
    assert L.dwt_hip_compare(0, b.ptr, b.ptr, w * 4, 4, w, h) == 1
    a.upload(ii)
    b.upload(ii)
    assert L.dwt_hip_compare(1, a.ptr, b.ptr, w * 4, 4, w, h) == 0
    ii[299, 499] += 1
    b.upload(ii)
    assert L.dwt_hip_compare(1, a.ptr, b.ptr, w * 4, 4, w, h) == 1
    a.free()
    b.free()


def test_finish_releases_and_context_stays_usable(dwt, oracle):
    """dwt_util_finish frees the workspace (src/libdwt.c:19186 is the release hook); a later
    call re-allocates it.  init is idempotent."""
    img = np.random.default_rng(9).random((256, 256), dtype=np.float32)
    want = img.copy()
    oracle.fwd("cdf97_2f_s", want, 3)
    for _ in range(2):
        dwt.dwt_util_init()
        got = img.copy()
        dwt.dwt_cdf97_2f_s(got, 1024, 4, 256, 256, 256, 256, 3)
        assert np.array_equal(bits(got), bits(want))
        d = dwt.DeviceImage(256, 256).upload(img)
        dwt.dwt_cdf97_2f_s(d.ptr, 1024, 4, 256, 256, 256, 256, 3)
        assert np.array_equal(bits(d.download(np.float32)), bits(want))
        d.free()
        dwt.dwt_util_finish()
    dwt.dwt_util_init()


def test_extreme_shapes_fused_equals_line_passes():
    """Maximum sizes (SURVEY s8c edge cases): 32768^2 (4 GiB), 64 x 2^20, 2^20 x 64, odd
    70001-wide rows, 12 levels, the interleaved layout, and 3-D volumes up to 1024^3 (fused
    one-pass level vs two passes): the fused kernels equal the exact / two-pass ones bit for
    bit and the round trip closes.  Own process (torch supplies the device tensors)."""
    import os
    import subprocess
    import sys

    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "large_sanity.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if "round-trip" in l]
    assert len(lines) == 14, out.stdout
    for l in lines:
        if "interleaved" in l:  # round 4: fused, generic and in-place calls, forward and inverse, bit for bit
            assert "forward fused == line passes == in place: True" in l and "inverse alike: True" in l, l
        elif "in place" in l:  # round 3: the one-pass in-place 3-D levels at 1024^3 and on a ragged volume
            assert "one-pass forward == out of place: True" in l and "one-pass inverse == two-pass: True" in l, l
        else:
            assert "fused == line passes: True" in l, l
        err = float(l.rsplit(" ", 1)[1])
        assert err == 0.0 if "cdf53_i" in l else err < 1e-5, l


@pytest.mark.parametrize("wname", ["cdf53_i", "cdf97_i"])
@pytest.mark.parametrize("shape", [(280, 1), (1, 300), (4, 1), (1, 2)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_int_single_sample_lines_out_of_place(dwt, oracle, wname, shape):
    """The int kernels leave a lone sample as it is (src/libdwt.c:10961); with distinct source and
    destination the samples still have to arrive in the destination (found by scripts/stress.py)."""
    ff, fi, dt = NAMES[wname]
    h, w = shape
    img = rand_img(np.random.default_rng(h * 31 + w), h, w, dt)
    want = img.copy()
    jw = oracle.fwd(ff, want, -1, decompose_one=1)
    pitch = ((w * 4 + 63) // 64) * 64
    pad = np.zeros((h, pitch // 4), dt)
    pad[:, :w] = img
    a = dwt.DeviceImage(h, w, 4, pitch).upload(pad)
    b = dwt.DeviceImage(h, w, 4, pitch).upload(np.full_like(pad, 77))
    j = dwt._fwd(dwt.WAVELET_ID[wname], a.ptr, b.ptr, pitch, 4, w, h, w, h, -1, 1, 0, "fwd")
    assert j == jw
    assert np.array_equal(b.download(dt)[:, :w], want)
    dwt._inv(dwt.WAVELET_ID[wname], b.ptr, a.ptr, pitch, 4, w, h, w, h, j, 1, 0, "inv")
    assert np.array_equal(a.download(dt)[:, :w], img)
    a.free()
    b.free()


@pytest.mark.parametrize("wname", ["cdf53_i", "cdf97_i"])
@pytest.mark.parametrize("shape", [(2, 2), (2, 3), (3, 5), (5, 4), (8, 8), (37, 53), (64, 65), (65, 64), (300, 513), (514, 301), (1100, 1300), (1027, 2050)],
                         ids=lambda s: f"{s[0]}x{s[1]}")
def test_int_wavelets_over_the_whole_int32_range(dwt, oracle, wname, shape):
    """Bit-exact for ANY int32 input, not only small ones: the reference's int kernels wrap modulo 2^32
    and the int 5/3 has line-end formulas of its own (`(d+1)>>1`, `-= s`: src/libdwt.c:10971-10976,
    11768-11773) that differ from the reflected interior form once the doubled term wraps
    (|x| >= 2^30).  Samples over the whole range with the wrap points forced onto the borders; odd and
    even sizes, single- and multi-tile; host entry (fused sweeps and exact line passes), device entry in
    place and out of place; one level and full depth; rows and columns; forward and inverse."""
    ff, fi, dt = NAMES[wname]
    h, w = shape
    rng = np.random.default_rng(h * 131 + w)
    wid = dwt.WAVELET_ID[wname]
    for j_in in (1, -1):
        img = full_range_ints(rng, (h, w))
        want = img.copy()
        jw = oracle.fwd(ff, want, j_in)
        rec_want = want.copy()
        oracle.inv(fi, rec_want, jw)
        for accel in (0, 1):
            dwt.dwt_util_set_accel(accel)
            try:
                got = img.copy()
                assert dwt.FORWARD[wname](got, got.strides[0], 4, w, h, w, h, j_in) == jw
                assert np.array_equal(got, want), f"forward, host entry, accel {accel}, j {j_in}"
                dwt.INVERSE[wname](got, got.strides[0], 4, w, h, w, h, jw)
                assert np.array_equal(got, rec_want), f"inverse, host entry, accel {accel}, j {j_in}"
            finally:
                dwt.dwt_util_set_accel(0)
        for inplace in (True, False):
            a = dwt.DeviceImage(h, w).upload(img)
            b = a if inplace else dwt.DeviceImage(h, w).upload(np.zeros_like(img))
            assert dwt._fwd(wid, a.ptr, b.ptr, a.stride_x, 4, w, h, w, h, j_in, 0, 0, "fwd") == jw
            assert np.array_equal(b.download(dt), want), f"forward, device entry, inplace {inplace}, j {j_in}"
            dwt._inv(wid, b.ptr, a.ptr, a.stride_x, 4, w, h, w, h, jw, 0, 0, "inv")
            assert np.array_equal(a.download(dt), rec_want), f"inverse, device entry, inplace {inplace}, j {j_in}"
            a.free()
            if b is not a:
                b.free()
    if wname == "cdf53_i":
        assert np.array_equal(rec_want, img), "the int 5/3 is reversible over the whole range"


def test_int_tile_variants_agree_over_the_whole_int32_range(dwt, oracle):
    """The line ends fall into different lanes / tiles / ring slots with every tile geometry."""
    h, w = 515, 1030
    img = full_range_ints(np.random.default_rng(99), (h, w))
    want = img.copy()
    jw = oracle.fwd("cdf53_2f_i", want, 3)
    wid = dwt.WAVELET_ID["cdf53_i"]
    try:
        for cpt in (4, 8):
            for tp in (2, 8, 64):
                for waves in (1, 4):
                    dwt.set_option("cpt", cpt)
                    dwt.set_option("tile_pairs", tp)
                    dwt.set_option("waves", waves)
                    a = dwt.DeviceImage(h, w).upload(img)
                    b = dwt.DeviceImage(h, w).upload(np.zeros_like(img))
                    assert dwt._fwd(wid, a.ptr, b.ptr, a.stride_x, 4, w, h, w, h, 3, 0, 0, "fwd") == jw
                    assert np.array_equal(b.download(np.int32), want), (cpt, tp, waves)
                    dwt._inv(wid, b.ptr, a.ptr, a.stride_x, 4, w, h, w, h, jw, 0, 0, "inv")
                    assert np.array_equal(a.download(np.int32), img), (cpt, tp, waves)
                    a.free()
                    b.free()
    finally:
        for k, v in (("cpt", 0), ("tile_pairs", 0), ("waves", 4)):
            dwt.set_option(k, v)


def test_placement_entries(dwt, oracle):
    """dwt_hip_alloc_batch (source + destination + LL scratch of a resident batch, the destination and the
    scratch chosen by timing), the library's own scratch search on the first large forward call, and a
    caller-owned workspace (dwt_hip_set_workspace): the transforms give the oracle's bits on all of them;
    a workspace that is too small is refused; two NULLs hand the scratch back."""
    import ctypes as C

    L = dwt.lib
    nb, h, w, J = 3, 520, 1030, 3
    rng = np.random.default_rng(41)
    imgs = rng.random((nb, h, w), dtype=np.float32)
    want = imgs.copy()
    for k in range(nb):
        oracle.fwd("cdf97_2f_s", want[k], J)
    try:
        dwt.set_option("place_min_mib", 0)   # search whatever the size
        dwt.set_option("place_tries", 3)
        dwt.dwt_util_finish()                # no scratch from earlier tests
        src, dst = dwt.alloc_batch("cdf97_s", nb, w, h, J)
        rep = dwt.alloc_batch_report()
        assert dwt.alloc_batch_note() == ""
        assert rep["arena_GiB"] >= 8 and rep["dst_positions_tried"] >= 1 and rep["scratch_positions_tried"] >= 1, rep
        assert 0 < rep["whole_call_ms_best_worst"][0] <= rep["whole_call_ms_best_worst"][1]
        assert L.dwt_hip_memcpy_h2d(src, imgs.ctypes.data, imgs.nbytes) == 0
        assert dwt.transform2d_batch("cdf97_s", 0, src, dst, h * w * 4, nb, w * 4, w, h, J) == J
        assert dwt.placement_report()[0] == [], "the scratch alloc_batch installed is in place: no second search"
        got = np.empty_like(imgs)
        assert L.dwt_hip_memcpy_d2h(got.ctypes.data, dst, got.nbytes) == 0
        assert np.array_equal(bits(got), bits(want))
        # a caller-owned workspace right after alloc_batch, no finish in between: the scratch the arena left with the
        # context is a mapped range that only the library's own release can free (round 4 called hipFree on it)
        b0 = nb * ((w + 1) // 2 + 3) * ((h + 1) // 2) * 4 + 64
        b1 = nb * ((w + 3) // 4 + 3) * ((h + 3) // 4) * 4 + 64
        w0, w1 = L.dwt_hip_malloc(b0), L.dwt_hip_malloc(b1)
        assert L.dwt_hip_set_workspace(w0, b0, w1, b1) == 0, dwt.last_error()
        zero_img = np.zeros_like(imgs)  # (bound to a name: a temporary's address would dangle)
        assert L.dwt_hip_memcpy_h2d(dst, zero_img.ctypes.data, imgs.nbytes) == 0
        assert dwt.transform2d_batch("cdf97_s", 0, src, dst, h * w * 4, nb, w * 4, w, h, J) == J
        assert L.dwt_hip_memcpy_d2h(got.ctypes.data, dst, got.nbytes) == 0
        assert np.array_equal(bits(got), bits(want))
        assert L.dwt_hip_set_workspace(None, 0, None, 0) == 0
        for p in (w0, w1):
            L.dwt_hip_free(p)
        # the library's own search is EXPLICIT (dwt_hip_tune): a fresh context, the caller's buffers; a plain
        # transform call never searches
        dwt.dwt_util_finish()
        assert dwt.transform2d_batch("cdf97_s", 0, src, dst, h * w * 4, nb, w * 4, w, h, J) == J
        assert dwt.placement_report()[0] == [], "no search inside a transform call"
        dwt.dwt_util_finish()
        dwt.tune("cdf97_s", 0, src, dst, h * w * 4, nb, w * 4, w, h, J)
        trials, kept = dwt.placement_report()
        assert len(trials) >= 1 and 0 <= kept < len(trials)
        zero_img = np.zeros_like(imgs)  # (bound to a name: a temporary's address would dangle)
        assert L.dwt_hip_memcpy_h2d(dst, zero_img.ctypes.data, imgs.nbytes) == 0
        assert dwt.transform2d_batch("cdf97_s", 0, src, dst, h * w * 4, nb, w * 4, w, h, J) == J
        assert L.dwt_hip_memcpy_d2h(got.ctypes.data, dst, got.nbytes) == 0
        assert np.array_equal(bits(got), bits(want))
        # ... unless the environment asks for it (DWT_HIP_TUNE=1 = option "tune_in_call"): programs that only know libdwt.h
        dwt.dwt_util_finish()
        dwt.set_option("tune_in_call", 1)
        assert dwt.transform2d_batch("cdf97_s", 0, src, dst, h * w * 4, nb, w * 4, w, h, J) == J
        dwt.set_option("tune_in_call", 0)
        assert len(dwt.placement_report()[0]) >= 1
        assert L.dwt_hip_memcpy_d2h(got.ctypes.data, dst, got.nbytes) == 0
        assert np.array_equal(bits(got), bits(want))
        # a batch too small for the search says so instead of hiding it
        dwt.set_option("place_min_mib", 1024)
        s2_, d2_ = dwt.alloc_batch("cdf97_s", nb, w, h, J)
        assert dwt.alloc_batch_report()["arena_GiB"] == 0 and "plain allocations" in dwt.alloc_batch_note()
        for p in (s2_, d2_):
            L.dwt_hip_free(p)
        dwt.set_option("place_min_mib", 0)
        # caller-owned workspace
        b0 = nb * ((w + 1) // 2 + 3) * ((h + 1) // 2) * 4 + 64
        b1 = nb * ((w + 3) // 4 + 3) * ((h + 3) // 4) * 4 + 64
        w0, w1 = L.dwt_hip_malloc(b0), L.dwt_hip_malloc(b1)
        assert L.dwt_hip_set_workspace(w0, b0, w1, b1) == 0, dwt.last_error()
        zero_img = np.zeros_like(imgs)  # (bound to a name: a temporary's address would dangle)
        assert L.dwt_hip_memcpy_h2d(dst, zero_img.ctypes.data, imgs.nbytes) == 0
        assert dwt.transform2d_batch("cdf97_s", 0, src, dst, h * w * 4, nb, w * 4, w, h, J) == J
        assert L.dwt_hip_memcpy_d2h(got.ctypes.data, dst, got.nbytes) == 0
        assert np.array_equal(bits(got), bits(want))
        assert L.dwt_hip_set_workspace(w0, 4096, w1, 4096) == 0
        with pytest.raises(dwt.DwtError):
            dwt.transform2d_batch("cdf97_s", 0, src, dst, h * w * 4, nb, w * 4, w, h, J)
        assert L.dwt_hip_set_workspace(None, 0, None, 0) == 0
        assert dwt.transform2d_batch("cdf97_s", 0, src, dst, h * w * 4, nb, w * 4, w, h, J) == J
        for p in (w0, w1, src, dst):
            L.dwt_hip_free(p)
    finally:
        L.dwt_hip_set_workspace(None, 0, None, 0)
        dwt.set_option("tune_in_call", 0)
        dwt.set_option("place_min_mib", 1024)
        dwt.set_option("place_tries", 4)


@pytest.mark.parametrize("wname", ["cdf97_s", "cdf53_i"])
def test_batch_sharded_over_devices_in_one_process(dwt, oracle, wname):
    """dwt_hip_transform2d_batch_sharded (one process, one host thread + context per device, image
    b -> slot b*G/B, peer copies for the split only): devices {0}, {0, 0} and {0, 0, 0} (several contexts on
    the one GPU of the test box -- the same code path as several GPUs) give the oracle's bits, forward and
    inverse, dense and padded pitch (bytes between the frames untouched), more slots than images."""
    ff, fi, dt = NAMES[wname]
    L = dwt.lib
    nb, h, w, J = 5, 260, 520, 3
    rng = np.random.default_rng(77)
    for pitch_e in (w, w + 12):
        buf = rand_img(rng, nb * h, pitch_e, dt).reshape(nb, h, pitch_e)
        want = buf.copy()
        for k in range(nb):
            oracle.fwd(ff, want[k][:, :w], J)
        rec = want.copy()
        for k in range(nb):
            oracle.inv(fi, rec[k][:, :w], J)
        src, dst = L.dwt_hip_malloc(buf.nbytes), L.dwt_hip_malloc(buf.nbytes)
        bs = h * pitch_e * 4
        for devices in ([0], [0, 0], [0, 0, 0], [0] * 7):
            fill = np.full_like(buf, 7)
            assert L.dwt_hip_memcpy_h2d(src, buf.ctypes.data, buf.nbytes) == 0 and L.dwt_hip_memcpy_h2d(dst, fill.ctypes.data, fill.nbytes) == 0
            assert dwt.transform2d_batch_sharded(wname, 0, src, dst, bs, nb, pitch_e * 4, w, h, J, devices) == J
            got = np.empty_like(buf)
            assert L.dwt_hip_memcpy_d2h(got.ctypes.data, dst, got.nbytes) == 0
            assert np.array_equal(bits(got[:, :, :w]), bits(want[:, :, :w])), (devices, pitch_e)
            assert np.all(got[:, :, w:] == 7), "bytes outside the frames keep their values"
            # inverse: coefficients in dst -> src
            assert dwt.transform2d_batch_sharded(wname, 1, dst, src, bs, nb, pitch_e * 4, w, h, J, devices) == J
            assert L.dwt_hip_memcpy_d2h(got.ctypes.data, src, got.nbytes) == 0
            assert np.array_equal(bits(got[:, :, :w]), bits(rec[:, :, :w])), (devices, pitch_e)
        with pytest.raises(dwt.DwtError):
            dwt.transform2d_batch_sharded(wname, 0, src, dst, bs, nb, pitch_e * 4, w, h, J, [0, 99])
        L.dwt_hip_free(src)
        L.dwt_hip_free(dst)


def test_measured_tile_heights_do_not_change_the_bits(dwt, oracle):
    """Large levels (an input of 512 MiB and more) get their tile height from dwt_hip_tune, which times 64 / 32 / 16 row pairs once
    per shape: a scheduling choice only -- the batch's coefficients are the oracle's with measured heights, with the
    launcher's rule (tune_tiles = 0) and with measured heights again; the transform calls themselves measure nothing."""
    L = dwt.lib
    nb, n, J = 33, 2048, 3   # level 0: 528 MiB
    imgs = np.random.default_rng(61).random((nb, n, n), dtype=np.float32)
    src, dst = L.dwt_hip_malloc(imgs.nbytes), L.dwt_hip_malloc(imgs.nbytes)
    assert L.dwt_hip_memcpy_h2d(src, imgs.ctypes.data, imgs.nbytes) == 0
    want0, want4 = imgs[0].copy(), imgs[nb - 1].copy()
    oracle.fwd("cdf97_2f_s", want0, J)
    oracle.fwd("cdf97_2f_s", want4, J)
    try:
        for tune in (1, 0, 1):
            dwt.set_option("tune_tiles", tune)
            assert dwt.get_option("tune_tiles") == tune
            dwt.dwt_util_finish()  # forget what was measured
            before = dwt.get_option("tile_cache_size")
            dwt.tune("cdf97_s", 0, src, dst, n * n * 4, nb, n * 4, n, n, J)
            assert dwt.get_option("tile_cache_size") - before == (1 if tune else 0)  # level 0 is the one level beyond the Infinity Cache
            zero_img = np.zeros_like(imgs)  # (bound to a name: a temporary's address would dangle)
            assert L.dwt_hip_memcpy_h2d(dst, zero_img.ctypes.data, imgs.nbytes) == 0
            launches = dwt.get_option("stat_launches")
            for _ in range(2):
                assert dwt.transform2d_batch("cdf97_s", 0, src, dst, n * n * 4, nb, n * 4, n, n, J) == J
            assert dwt.get_option("stat_launches") - launches == 2 * J, "a transform call is J launches, nothing else"
            got = np.empty_like(imgs)
            assert L.dwt_hip_memcpy_d2h(got.ctypes.data, dst, got.nbytes) == 0
            assert np.array_equal(bits(got[0]), bits(want0)) and np.array_equal(bits(got[nb - 1]), bits(want4)), tune
    finally:
        dwt.set_option("tune_tiles", 1)
        L.dwt_hip_free(src)
        L.dwt_hip_free(dst)


def test_mapped_buffers_are_ordinary_device_memory(dwt, oracle):
    """The diagnosis instruments (dwt_hip_malloc_mapped / _spread: buffers mapped from physical pieces
    through HIP's virtual-memory API) hand out memory every entry accepts and dwt_hip_free releases."""
    L = dwt.lib
    h, w = 300, 700
    img = np.random.default_rng(5).random((h, w), dtype=np.float32)
    want = img.copy()
    jw = oracle.fwd("cdf97_2f_s", want, 3)
    for p in (L.dwt_hip_malloc_mapped(img.nbytes, 2 << 20, 2, 0), L.dwt_hip_malloc_spread(img.nbytes, 2 << 20)):
        assert p, dwt.last_error()
        assert L.dwt_hip_is_device_pointer(p)
        assert L.dwt_hip_memcpy_h2d(p, img.ctypes.data, img.nbytes) == 0
        assert dwt.dwt_cdf97_2f_s(p, w * 4, 4, w, h, w, h, 3) == jw
        got = np.empty_like(img)
        assert L.dwt_hip_memcpy_d2h(got.ctypes.data, p, got.nbytes) == 0
        assert np.array_equal(bits(got), bits(want))
        L.dwt_hip_free(p)


def test_randomised_soak():
    """scripts/stress.py for 20 s: random shapes, levels, wavelets, entries and layouts; the fused
    kernels against the exact line-pass / two-pass kernels bit for bit, plus round trips."""
    import os
    import subprocess
    import sys

    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "stress.py"), "20", "7"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " 0 mismatches" in out.stdout, (out.stdout[-2000:], out.stderr[-1000:])


@pytest.mark.parametrize("wname", ["cdf97_s", "cdf53_i", "cdf53_s", "cdf97_i"])
def test_in_place_copy_rides_along_with_the_deeper_levels(dwt, oracle, wname):
    """In-place device calls on one image stage level 0's detail subbands; the copy that brings them back (forward) / moves
    them aside (inverse) is handed out block by block to the deeper levels' launches (option ride_copy, default on).  Same
    bits as the oracle with the copy riding along -- a lot per level, a little per level plus a remainder launch, nothing --
    for odd shapes, padded pitches and level counts from 1 up."""
    ff, fi, dt = NAMES[wname]
    rng = np.random.default_rng(31)
    try:
        for (h, w, pitch_e, J) in ((1030, 2100, 2100, 4), (2048, 4096, 4100, 5), (771, 517, 520, 3), (600, 900, 900, 1), (4096, 1024, 1024, -1)):
            img = rand_img(rng, h, w, dt)
            want = img.copy()
            jw = oracle.fwd(ff, want, J)
            rec = want.copy()
            oracle.inv(fi, rec, jw)
            for ride, mib in ((1, 32), (1, 1), (1, 4096), (0, 32)):
                dwt.set_option("ride_copy", ride)
                dwt.set_option("ride_mib", mib)
                d = dwt.DeviceImage(h, w, pitch_bytes=pitch_e * 4)
                buf = np.full((h, pitch_e), 9, dtype=dt)
                buf[:, :w] = img
                d.upload(buf)
                assert getattr(dwt, "dwt_" + ff)(d.ptr, pitch_e * 4, 4, w, h, w, h, J) == jw
                got = d.download(dt)
                assert np.array_equal(bits(got[:, :w]), bits(want)) and np.all(got[:, w:] == 9), (h, w, J, ride, mib)
                getattr(dwt, "dwt_" + fi)(d.ptr, pitch_e * 4, 4, w, h, w, h, jw)
                got = d.download(dt)
                assert np.array_equal(bits(got[:, :w]), bits(rec)) and np.all(got[:, w:] == 9), (h, w, J, ride, mib, "inverse")
                d.free()
    finally:
        dwt.set_option("ride_copy", 1)
        dwt.set_option("ride_mib", 32)


def test_calls_are_graph_capturable():
    """scripts/graph_replay.py: device-pointer calls captured into a HIP graph and replayed give the eager call's bits --
    a call is launches on the caller's stream and nothing else (no allocation, no synchronisation, no measurement)."""
    import os
    import subprocess
    import sys

    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "graph_replay.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and " 0 mismatches" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


@pytest.mark.parametrize("wname", ["cdf97_s", "cdf53_i"])
@pytest.mark.parametrize("shape,levels", [((4096, 8192), 5), ((8192, 8192), 5), ((5001, 4097), 3), ((8192, 2100), 1), ((2049, 8200), -1), ((16390, 1100), 2)],
                         ids=lambda v: str(v))
def test_host_pointer_call_pipelined_under_the_transfers(dwt, oracle, wname, shape, levels):
    """Host-pointer calls on images of 64 MiB and more run level 0 (forward: and level 1) band by band while the image is
    still crossing PCIe (the caller's memory pinned in place, uploads, kernels and downloads on three streams).  Same bits
    as the plain upload / transform / download path (host_pipeline = 0), forward and inverse, in place and out of place,
    odd sizes and padded rows; the round trip closes -- and both are the ORACLE's bits (forward coefficients and
    reconstruction), so the default drop-in path for large host images is pinned directly, not through its sibling."""
    h, w = shape
    ff, fi, dt = NAMES[wname]
    rng = np.random.default_rng(h + w)
    pitch = w + 24  # elements: rows with padding the call must neither read as image nor touch
    base = np.full((h, pitch), 7, dtype=dt)
    img = rng.integers(-30000, 30000, size=(h, w)).astype(dt) if dt == np.int32 else rng.random((h, w), dtype=np.float32)
    base[:, :w] = img
    outs = []
    for pipe in (0, 1, 1):
        dwt.set_option("host_pipeline", pipe)
        try:
            a = base.copy()
            j1 = getattr(dwt, "dwt_" + ff)(a, pitch * 4, 4, w, h, w, h, levels)  # in place
            src = base.copy()
            b = np.full((h, pitch), 9, dtype=dt)
            j2 = dwt._fwd(dwt.WAVELET_ID[wname], src, b, pitch * 4, 4, w, h, w, h, levels, 0, 0, "forward, out of place")
        finally:
            dwt.set_option("host_pipeline", 1)
        assert j1 == j2 and np.array_equal(src, base)
        assert np.all(a[:, w:] == 7) and np.all(b[:, w:] == 9)
        assert np.array_equal(bits(a[:, :w]), bits(b[:, :w]))
        fwd = a[:, :w].copy()
        # and back: in place on `a`, out of place from `b` into `c`
        dwt.set_option("host_pipeline", pipe)
        try:
            getattr(dwt, "dwt_" + fi)(a, pitch * 4, 4, w, h, w, h, j1)
            keep = b.copy()
            c = np.full((h, pitch), 5, dtype=dt)
            dwt._inv(dwt.WAVELET_ID[wname], b, c, pitch * 4, 4, w, h, w, h, j1, 0, 0, "inverse, out of place")
        finally:
            dwt.set_option("host_pipeline", 1)
        assert np.array_equal(b, keep) and np.all(a[:, w:] == 7) and np.all(c[:, w:] == 5)
        assert np.array_equal(bits(a[:, :w]), bits(c[:, :w]))
        outs.append((j1, fwd, a[:, :w].copy()))
    assert outs[0][0] == outs[1][0] == outs[2][0]
    for k in (1, 2):
        assert np.array_equal(bits(outs[0][k]), bits(outs[1][k])) and np.array_equal(bits(outs[0][k]), bits(outs[2][k]))
    rec = outs[0][2]
    assert np.array_equal(rec, img) if dt == np.int32 else np.abs(rec - img).max() < 1e-4
    want = img.copy()
    jw = oracle.fwd(ff, want, levels)
    assert jw == outs[1][0] and np.array_equal(bits(outs[1][1]), bits(want)), "pipelined forward == oracle"
    oracle.inv(fi, want, jw)
    assert np.array_equal(bits(outs[1][2]), bits(want)), "pipelined inverse == oracle"


def test_host_pointer_call_from_a_read_only_mapping(dwt, oracle, tmp_path):
    """`_s2` with a source the library cannot pin for writing -- a read-only file mapping (the caller's memory is
    pinned in place with hipHostRegister for the pipelined path; a mapping that refuses takes the plain path): the call
    succeeds either way and gives the oracle's bits; the source is untouched."""
    import mmap

    h, w, J = 4096, 8192, 4   # 128 MiB: above the pipelined path's threshold
    img = np.random.default_rng(12).random((h, w), dtype=np.float32)
    path = tmp_path / "src.raw"
    img.tofile(path)
    want = img.copy()
    jw = oracle.fwd("cdf97_2f_s", want, J)
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        src = np.frombuffer(mm, dtype=np.float32).reshape(h, w)
        assert not src.flags.writeable
        dst = np.zeros((h, w), dtype=np.float32)
        assert dwt._fwd(dwt.CDF97_S, src.ctypes.data, dst, w * 4, 4, w, h, w, h, J, 0, 0, "dwt_cdf97_2f_s2, read-only source") == jw
        assert np.array_equal(bits(dst), bits(want))
        assert np.array_equal(bits(src), bits(img))
        rec = np.zeros((h, w), dtype=np.float32)
        dwt._inv(dwt.CDF97_S, dst, rec, w * 4, 4, w, h, w, h, jw, 0, 0, "dwt_cdf97_2i_s2")
        oracle.inv("cdf97_2i_s", want, jw)
        assert np.array_equal(bits(rec), bits(want))
        del src
        mm.close()


def test_two_threads_on_the_halves_of_one_allocation(dwt, oracle):
    """Two host threads (each with its own context) transform the two halves of ONE allocation at the same time through
    the host-pointer entries: the halves share the page at their boundary, so the pinning of one call meets the pinning
    of the other (whichever loses takes the plain path).  Both halves give the oracle's bits, forward and inverse."""
    import threading

    h, w, J = 2304, 8200, 3   # 72 MiB per half; a row is not a multiple of the page size
    buf = np.random.default_rng(99).random((2 * h, w), dtype=np.float32)
    orig = buf.copy()
    want = buf.copy()
    for k in range(2):
        oracle.fwd("cdf97_2f_s", want[k * h:(k + 1) * h], J)
    errs = []

    def run(k, inverse):
        try:
            half = buf[k * h:(k + 1) * h]
            if inverse:
                dwt.dwt_cdf97_2i_s(half.ctypes.data, w * 4, 4, w, h, w, h, J)
            else:
                assert dwt.dwt_cdf97_2f_s(half.ctypes.data, w * 4, 4, w, h, w, h, J) == J
        except Exception as e:  # noqa: BLE001
            errs.append((k, inverse, repr(e)))
        finally:
            dwt.dwt_util_finish()  # the thread's own context

    for inverse in (False, True):
        th = [threading.Thread(target=run, args=(k, inverse)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        if not inverse:
            assert np.array_equal(bits(buf), bits(want))
            for k in range(2):
                oracle.inv("cdf97_2i_s", want[k * h:(k + 1) * h], J)
        else:
            assert np.array_equal(bits(buf), bits(want)) and np.abs(buf - orig).max() < 1e-4
