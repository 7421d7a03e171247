"""ctypes bindings for the test-only checkers under oracle/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  `Oracle` wraps oracle/_build/libdwt_oracle.so (the CPU restatement) and
`Reference` wraps oracle/_ref/libdwt_ref.so (the reference compiled from its own
sources by oracle/Makefile, present only where it was built).  Both expose the same
numpy-level helpers so a test can swap one for the other.
"""
import contextlib
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# (DWT_ORACLE_SO: another build of the same restatement -- the sanitizer build of tests/san/Makefile)
ORACLE_SO = os.environ.get("DWT_ORACLE_SO") or os.path.join(ORACLE_DIR, "_build", "libdwt_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libdwt_ref.so")
REFERENCE_SRC = "/root/reference"

_I = C.c_int
_P = C.c_void_p
_FWD = [_P, _I, _I, _I, _I, _I, _I, C.POINTER(_I), _I, _I]
_INV = [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I]
_FWD2 = [_P, _P, _I, _I, _I, _I, _I, _I, C.POINTER(_I), _I, _I]
_INV2 = [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I]

# entry -> (ctypes signature, numpy dtype)
ENTRIES = {
    "cdf97_2f_s": (_FWD, np.float32),
    "cdf97_2i_s": (_INV, np.float32),
    "cdf97_2f_s2": (_FWD2, np.float32),
    "cdf97_2i_s2": (_INV2, np.float32),
    "cdf53_2f_i": (_FWD, np.int32),
    "cdf53_2i_i": (_INV, np.int32),
    "cdf53_2f_s": (_FWD, np.float32),
    "cdf53_2i_s": (_INV, np.float32),
    "cdf97_2f_d": (_FWD, np.float64),
    "cdf97_2i_d": (_INV, np.float64),
    "cdf53_2f_d": (_FWD, np.float64),
    "cdf53_2i_d": (_INV, np.float64),
    "cdf97_2f_i": (_FWD, np.int32),
    "cdf97_2i_i": (_INV, np.int32),
    # interleaved (in-place lifting) layout
    "cdf97_2f_inplace_s": (_FWD, np.float32),
    "cdf97_2i_inplace_s": (_INV, np.float32),
    "cdf53_2f_inplace_s": (_FWD, np.float32),
    "cdf53_2i_inplace_s": (_INV, np.float32),
    "cdf97_2f_inplace_i": (_FWD, np.int32),
    "cdf97_2i_inplace_i": (_INV, np.int32),
}

# dwt-simple.h entries: (ptr, size_x, size_y, stride_x, stride_y, int *j, decompose_one)
_NEW = [_P, _I, _I, _I, _I, C.POINTER(_I), _I]


def build_oracle(force=False):
    """Compile the restatement (and the reference when its sources are present)."""
    if force or not os.path.exists(ORACLE_SO) or (
        os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "dwt_oracle.c"))
    ):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])
    if os.path.isdir(os.path.join(REFERENCE_SRC, "src")) and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


class _Lib:
    prefix = ""

    def __init__(self, path):
        self.path = path
        self.lib = C.CDLL(path)
        for name, (sig, _) in ENTRIES.items():
            fn = getattr(self.lib, self.prefix + name)
            fn.argtypes = sig
            fn.restype = None

    # ---- 2-D transforms on numpy arrays (dense or padded pitch) ----
    def _call(self, name, img, j, size_o=None, size_i=None, decompose_one=0, zero_padding=0):
        """In-place transform of the 2-D array `img` (rows = y).  `size_o`/`size_i`
        are (x, y) pairs and default to the array shape.  Returns the level count
        the callee reports (forward) or `j` (inverse)."""
        sig, dt = ENTRIES[name]
        es = np.dtype(dt).itemsize
        assert img.dtype == dt and img.ndim == 2 and img.strides[1] == es
        h, w = img.shape
        sox, soy = size_o if size_o else (w, h)
        six, siy = size_i if size_i else (sox, soy)
        fn = getattr(self.lib, self.prefix + name)
        if "2f" in name:
            jj = _I(j)
            fn(img.ctypes.data, img.strides[0], es, sox, soy, six, siy, C.byref(jj), decompose_one, zero_padding)
            return jj.value
        fn(img.ctypes.data, img.strides[0], es, sox, soy, six, siy, j, decompose_one, zero_padding)
        return j

    def call_channel(self, name, img, channel, j, size_i=None, decompose_one=0, zero_padding=0):
        """One channel of an interleaved multi-channel image `img` (H, W, C), in place, the way
        the reference's OpenCV wrapper calls the entry (src/cvdwt.cpp:98-135): ptr = data +
        elemSize1*channel, stride_x = step, stride_y = elemSize = C*elemSize1."""
        sig, dt = ENTRIES[name]
        es = np.dtype(dt).itemsize
        assert img.dtype == dt and img.ndim == 3 and img.strides[2] == es and img.strides[1] == es * img.shape[2]
        h, w, _ = img.shape
        six, siy = size_i if size_i else (w, h)
        fn = getattr(self.lib, self.prefix + name)
        ptr = img.ctypes.data + es * channel
        if "2f" in name:
            jj = _I(j)
            fn(ptr, img.strides[0], img.strides[1], w, h, six, siy, C.byref(jj), decompose_one, zero_padding)
            return jj.value
        fn(ptr, img.strides[0], img.strides[1], w, h, six, siy, j, decompose_one, zero_padding)
        return j

    def fwd(self, name, img, j=-1, **kw):
        return self._call(name, img, j, **kw)

    def inv(self, name, img, j=-1, **kw):
        return self._call(name, img, j, **kw)

    def fdwt2(self, wavelet, img, j=-1, decompose_one=0, schedule="horizontal"):
        """dwt-simple.h forward transform, interleaved layout, in place; returns j."""
        assert img.dtype == np.float32 and img.ndim == 2 and img.strides[1] == 4
        fn = self._fdwt2_fn(wavelet, schedule)
        fn.argtypes = _NEW
        fn.restype = None
        jj = _I(j)
        fn(img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], 4, C.byref(jj), decompose_one)
        return jj.value

    def call2(self, name, src, dst, j, size_o=None, size_i=None, decompose_one=0, zero_padding=0):
        """Out-of-place `_s2` entries; src and dst share the pitch of `dst`."""
        assert src.strides == dst.strides and src.dtype == dst.dtype == np.float32
        h, w = dst.shape
        sox, soy = size_o if size_o else (w, h)
        six, siy = size_i if size_i else (sox, soy)
        fn = getattr(self.lib, self.prefix + name)
        if "2f" in name:
            jj = _I(j)
            fn(src.ctypes.data, dst.ctypes.data, dst.strides[0], 4, sox, soy, six, siy, C.byref(jj), decompose_one, zero_padding)
            return jj.value
        fn(src.ctypes.data, dst.ctypes.data, dst.strides[0], 4, sox, soy, six, siy, j, decompose_one, zero_padding)
        return j


class Oracle(_Lib):
    prefix = "oracle_"

    def __init__(self):
        build_oracle()
        super().__init__(ORACLE_SO)
        L = self.lib
        for n, t in (("line_cdf97_f_s", np.float32), ("line_cdf97_i_s", np.float32),
                     ("line_cdf53_f_i", np.int32), ("line_cdf53_i_i", np.int32),
                     ("line_cdf53_f_s", np.float32), ("line_cdf53_i_s", np.float32)):
            getattr(L, "oracle_" + n).argtypes = [_P, _I]
            getattr(L, "oracle_" + n).restype = None
        for n in ("oracle_cdf97_3f_s", "oracle_cdf97_3i_s"):
            getattr(L, n).argtypes = [_P, C.c_long, C.c_long, C.c_long, _I, _I, _I]
            getattr(L, n).restype = None
        for n in ("oracle_test_image_fill_s", "oracle_test_image_fill_i"):
            getattr(L, n).argtypes = [_P, _I, _I, _I, _I, _I]
            getattr(L, n).restype = None
        L.oracle_set_threads.argtypes = [_I]
        L.oracle_set_end_form.argtypes = [_I]
        L.oracle_max_threads.restype = _I
        # a GPU box exposes all host cores but only a share of them is ours: OpenMP with
        # hundreds of threads on a busy host is slower than 16
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        L.oracle_set_threads(max(1, min(16, avail)))

    def _fdwt2_fn(self, wavelet, schedule):
        return getattr(self.lib, "oracle_fdwt2_%s_s" % wavelet)

    def line(self, name, a):
        getattr(self.lib, "oracle_line_" + name)(a.ctypes.data, a.shape[0])
        return a

    def vol(self, name, v):
        """3-D single level in place on a C-contiguous (z, y, x) float32 array."""
        assert v.dtype == np.float32 and v.ndim == 3
        nz, ny, nx = v.shape
        getattr(self.lib, "oracle_" + name)(v.ctypes.data, v.strides[2], v.strides[1], v.strides[0], nx, ny, nz)
        return v

    def fill_s(self, img, rnd=0):
        self.lib.oracle_test_image_fill_s(img.ctypes.data, img.strides[0], 4, img.shape[1], img.shape[0], rnd)
        return img

    def fill_i(self, img, rnd=0):
        self.lib.oracle_test_image_fill_i(img.ctypes.data, img.strides[0], 4, img.shape[1], img.shape[0], rnd)
        return img

    def set_threads(self, n):
        self.lib.oracle_set_threads(n)

    @contextlib.contextmanager
    def reflected_ends(self):
        """Within the block the float 9/7 line ends are evaluated as c*(x+x) -- what whole-sample
        reflection of the taps gives and the HIP kernels compute -- instead of the reference's
        (2c)*x.  Same bits unless x+x overflows (oracle/dwt_oracle.c header)."""
        self.lib.oracle_set_end_form(0)
        try:
            yield self
        finally:
            self.lib.oracle_set_end_form(1)


class Reference(_Lib):
    """The reference library itself (libdwt 2015-02-18-dev), where it was built."""
    prefix = "dwt_"

    def __init__(self):
        if not os.path.exists(REF_SO):
            build_oracle()
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO)
        super().__init__(REF_SO)
        L = self.lib
        L.dwt_util_set_accel.argtypes = [_I]
        L.dwt_util_set_num_workers.argtypes = [_I]
        L.dwt_util_set_num_threads.argtypes = [_I]
        L.dwt_util_get_num_threads.restype = _I
        for n in ("dwt_util_test_image_fill_s", "dwt_util_test_image_fill_i"):
            getattr(L, n).argtypes = [_P, _I, _I, _I, _I, _I]
            getattr(L, n).restype = None
        L.dwt_util_get_opt_stride.argtypes = [_I]
        L.dwt_util_get_opt_stride.restype = _I
        L.dwt_util_get_stride.argtypes = [_I, _I]
        L.dwt_util_get_stride.restype = _I

    def _fdwt2_fn(self, wavelet, schedule):
        return getattr(self.lib, "fdwt2_%s_%s_s" % (wavelet, schedule))

    def fill_s(self, img, rnd=0):
        self.lib.dwt_util_test_image_fill_s(img.ctypes.data, img.strides[0], 4, img.shape[1], img.shape[0], rnd)
        return img

    def fill_i(self, img, rnd=0):
        self.lib.dwt_util_test_image_fill_i(img.ctypes.data, img.strides[0], 4, img.shape[1], img.shape[0], rnd)
        return img


HYBRID_DIR = os.path.join(ORACLE_DIR, "_ref", "hybrid")
HYBRID_SO = os.path.join(HYBRID_DIR, "libdwt_hybrid.so")


class Hybrid(Reference):
    """The reference compiled WITH the maintainer's dispatch lines (integration/libdwt_hip_dispatch.patch, built by
    `make -C oracle ref_hybrid`) and linked against the product library: accel 0 is libdwt's own CPU path, accel 100
    the MI355X backend -- in one binary."""

    def __init__(self):
        if not os.path.exists(HYBRID_SO) and os.path.isdir(os.path.join(REFERENCE_SRC, "src")):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref_hybrid"])
        if not os.path.exists(HYBRID_SO):
            raise FileNotFoundError(HYBRID_SO)
        _Lib.__init__(self, HYBRID_SO)
        L = self.lib
        L.dwt_util_set_accel.argtypes = [_I]
        L.dwt_util_set_num_workers.argtypes = [_I]
        L.dwt_util_set_num_threads.argtypes = [_I]


def have_reference():
    return os.path.exists(REF_SO) or os.path.isdir(os.path.join(REFERENCE_SRC, "src"))


def padded(h, w, dtype, pitch_elems=None, fill=None, rng=None):
    """A (h, w) view into a wider buffer so the row pitch differs from w*4."""
    pe = pitch_elems or w
    buf = np.zeros((h, pe), dtype=dtype)
    if rng is not None:
        if np.issubdtype(dtype, np.floating):
            buf[:] = rng.random((h, pe), dtype=np.float32)
        else:
            buf[:] = rng.integers(-32768, 32768, size=(h, pe), dtype=np.int32)
    elif fill is not None:
        buf[:] = fill
    return buf, buf[:, :w]
