"""GPU: element-strided images RESIDENT ON THE DEVICE.  libdwt's transforms take
any element stride -- every line goes through dwt_util_memcpy_stride_s / _i (src/system.c:102-164) --
and the reference's OpenCV wrapper transforms one channel of an interleaved multi-channel matrix that
way (ptr = data + elemSize1*channel, stride_x = step, stride_y = elemSize: src/cvdwt.cpp:98-135).
Here the matrix lies in HBM: the frame is packed, transformed and spread back on the device
(libdwt_amd/csrc/dwt_strided.hip); bits equal to the reference's, the other channels and the pitch
padding untouched."""
import numpy as np
import pytest

from conftest import bits, full_range_ints, multichannel_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dwt():
    import libdwt_amd as d

    d.dwt_util_init()
    yield d
    d.dwt_util_finish()


class DevBuf:
    def __init__(self, dwt, arr):
        self.dwt, self.shape, self.dtype = dwt, arr.shape, arr.dtype
        a = np.ascontiguousarray(arr)
        self.nbytes = a.nbytes
        self.ptr = dwt.lib.dwt_hip_malloc(max(16, a.nbytes))
        assert self.ptr and dwt.lib.dwt_hip_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes) == 0

    def get(self):
        out = np.empty(self.shape, self.dtype)
        assert self.dwt.lib.dwt_hip_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes) == 0
        return out

    def free(self):
        self.dwt.lib.dwt_hip_free(self.ptr)


@pytest.mark.parametrize("case", multichannel_cases(), ids=lambda c: c[0]["name"])
def test_golden_multichannel_device_resident(dwt, case):
    """The 21 reference-generated multichannel fixtures with the matrix in HBM."""
    meta, src, fwd, inv = case
    wname = meta["wavelet"]
    es = src.dtype.itemsize
    (sox, soy), (six, siy) = meta["size_o"], meta["size_i"]
    d = DevBuf(dwt, src)
    ptr = d.ptr + es * meta["channel"]
    j = dwt.FORWARD[wname](ptr, src.strides[0], src.strides[1], sox, soy, six, siy, meta["j_in"],
                           meta["decompose_one"], meta["zero_padding"])
    assert j == meta["j_out"]
    assert np.array_equal(bits(d.get()), bits(fwd)), "forward differs from the reference's coefficients"
    dwt.INVERSE[wname](ptr, src.strides[0], src.strides[1], sox, soy, six, siy, j, meta["decompose_one"], meta["zero_padding"])
    assert np.array_equal(bits(d.get()), bits(inv)), "inverse differs from the reference's output"
    d.free()


def test_multichannel_all_channels_like_cv_dwt_transform_device_resident(dwt, oracle):
    """dwt::transform of the wrapper (src/cvdwt.cpp:303-350) loops over the channels of one Mat: a
    3-channel 1920x1080 float matrix in HBM, every channel forward in place through the C entry;
    equals the oracle channel by channel; the inverse restores the image."""
    rng = np.random.default_rng(8)
    h, w, c = 1080, 1920, 3
    img = rng.random((h, w, c), dtype=np.float32)
    want = img.copy()
    d = DevBuf(dwt, img)
    for ch in range(c):
        jw = oracle.call_channel("cdf97_2f_s", want, ch, 4)
        jg = dwt.dwt_cdf97_2f_s(d.ptr + 4 * ch, img.strides[0], img.strides[1], w, h, w, h, 4)
        assert jg == jw == 4
        got = d.get()
        assert np.array_equal(bits(got[:, :, :ch + 1]), bits(want[:, :, :ch + 1]))
        assert np.array_equal(bits(got[:, :, ch + 1:]), bits(img[:, :, ch + 1:])), "a channel not yet transformed changed"
    for ch in range(c):
        dwt.dwt_cdf97_2i_s(d.ptr + 4 * ch, img.strides[0], img.strides[1], w, h, w, h, 4)
    assert np.abs(d.get() - img).max() < 1e-5
    d.free()


@pytest.mark.parametrize("wname,dt", [("cdf53_i", np.int32), ("cdf97_d", np.float64), ("cdf53_s", np.float32)])
def test_s2_sparse_and_unaligned_strides_device_resident(dwt, oracle, wname, dt):
    """Out-of-place between two strided device images (the `_s2` shape of the call through
    dwt_hip_transform2d), a sparse frame with zero padding, int over the whole range, double; and
    byte strides that are not multiples of the element size (elements moved byte by byte)."""
    rng = np.random.default_rng(11)
    es = np.dtype(dt).itemsize
    h, w, c = 70, 150, 2
    if dt == np.int32:
        img = full_range_ints(rng, (h, w * c)).reshape(h, w, c)
    else:
        img = (rng.random((h, w, c)) * 2 - 1).astype(dt)
    ff = {"cdf53_i": "cdf53_2f_i", "cdf97_d": "cdf97_2f_d", "cdf53_s": "cdf53_2f_s"}[wname]
    fi = ff.replace("2f", "2i")
    want = img.copy()
    kw = dict(size_i=(120, 50), zero_padding=1)
    jw = oracle.call_channel(ff, want, 1, 3, **kw)
    d = DevBuf(dwt, img)
    jg = dwt.FORWARD[wname](d.ptr + es, img.strides[0], img.strides[1], w, h, 120, 50, 3, 0, 1)
    assert jg == jw and np.array_equal(bits(d.get()), bits(want))
    oracle.call_channel(fi, want, 1, jw, **kw)
    dwt.INVERSE[wname](d.ptr + es, img.strides[0], img.strides[1], w, h, 120, 50, jg, 0, 1)
    assert np.array_equal(bits(d.get()), bits(want))
    d.free()
    # unaligned: a byte buffer, elements at odd addresses, odd row pitch
    hh, ww = 37, 53
    sy, sx, off = es + 3, (es + 3) * ww + 5, 1
    raw = rng.integers(0, 256, size=sx * hh + off + 16, dtype=np.uint8)
    dense = img[:hh, :ww, 0].copy()
    view = np.lib.stride_tricks.as_strided(raw[off:].view(np.uint8), shape=(hh, ww, es), strides=(sx, sy, 1))
    view[:] = dense.view(np.uint8).reshape(hh, ww, es)
    d = DevBuf(dwt, raw)
    want2 = dense.copy()
    jw = oracle.fwd(ff, want2, -1)
    assert dwt.FORWARD[wname](d.ptr + off, sx, sy, ww, hh, ww, hh, -1) == jw
    got = d.get()
    gview = np.lib.stride_tricks.as_strided(got[off:], shape=(hh, ww, es), strides=(sx, sy, 1))
    assert np.array_equal(np.ascontiguousarray(gview).view(dt).reshape(hh, ww).view(np.uint8), want2.view(np.uint8))
    expect = raw.copy()
    np.lib.stride_tricks.as_strided(expect[off:], shape=(hh, ww, es), strides=(sx, sy, 1))[:] = want2.view(np.uint8).reshape(hh, ww, es)
    assert np.array_equal(got, expect), "bytes between the elements changed"
    d.free()


def test_strided_device_images_reject_overlapping_rows(dwt):
    d = dwt.lib.dwt_hip_malloc(1 << 16)
    with pytest.raises(dwt.DwtError):
        dwt.dwt_cdf97_2f_s(d, 64, 12, 32, 8, 32, 8, 1)  # rows of 32 elements 12 bytes apart need 376 bytes
    with pytest.raises(dwt.DwtError):
        dwt.dwt_cdf97_2f_s(d, 512, 2, 32, 8, 32, 8, 1)  # elements overlap
    dwt.lib.dwt_hip_free(d)


def test_interleaved_layout_entries_on_a_strided_device_image(dwt, oracle):
    """dwt_cdf97_2f_inplace_s / _2i_inplace_s and fdwt2_cdf53 on one channel of a 2-channel matrix in HBM."""
    rng = np.random.default_rng(12)
    h, w, c = 130, 200, 2
    img = rng.random((h, w, c), dtype=np.float32)
    d = DevBuf(dwt, img)
    want = np.ascontiguousarray(img[:, :, 1])
    jw = oracle.fwd("cdf97_2f_inplace_s", want, 3)
    assert dwt.dwt_cdf97_2f_inplace_s(d.ptr + 4, img.strides[0], img.strides[1], w, h, w, h, 3) == jw
    got = d.get()
    assert np.array_equal(bits(got[:, :, 1]), bits(want)) and np.array_equal(bits(got[:, :, 0]), bits(img[:, :, 0]))
    oracle.inv("cdf97_2i_inplace_s", want, jw)
    dwt.dwt_cdf97_2i_inplace_s(d.ptr + 4, img.strides[0], img.strides[1], w, h, w, h, jw)
    got = d.get()
    assert np.array_equal(bits(got[:, :, 1]), bits(want)) and np.array_equal(bits(got[:, :, 0]), bits(img[:, :, 0]))
    d.free()
