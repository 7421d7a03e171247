"""libdwt_amd -- Python mirror of libdwt's 2-D DWT entry points over the MI355X backend.

The product is the C-ABI shared library ``libdwt_amd/libdwt_hip.so`` (C host code +
hand-written gfx950 HIP kernels; headers in ``include/``).  This module is only a
ctypes binding with the reference's function names and argument order
(``src/libdwt.h:562-573`` etc.), so tests read like programs written against libdwt:

    import libdwt_amd as dwt
    dwt.dwt_util_init()
    j = dwt.dwt_cdf97_2f_s(img, stride_x, 4, w, h, w, h, -1, 0, 0)   # returns levels done
    dwt.dwt_cdf97_2i_s(img, stride_x, 4, w, h, w, h, j, 0, 0)

``img`` may be a numpy array (host memory: staged through HBM), a torch tensor
(host or device), or a raw address (``int``) -- e.g. from ``dwt_hip_malloc``.

There is no CPU fallback anywhere: if the library is missing, importing this module
raises; if no gfx950 device is usable, every transform raises ``DwtError``.

When PyTorch is used in the same process (device tensors, streams, HIP graphs), import
``torch`` BEFORE this module: the library then binds to the HIP runtime torch ships
instead of loading a second one, and tensors' ``data_ptr()`` are valid for it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DWT_HIP_LIB") or os.path.join(_HERE, "libdwt_hip.so")  # (DWT_HIP_LIB: another build of the library, for A/B scripts)

CDF97_S, CDF53_I, CDF53_S, CDF97_D, CDF53_D, CDF97_I = 0, 1, 2, 3, 4, 5


class DwtError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C libdwt_amd/csrc` (hipcc, gfx950). libdwt_amd has no CPU fallback."
    )

lib = C.CDLL(LIB_PATH)

_I, _P, _S = C.c_int, C.c_void_p, C.c_size_t
_FWD = [_P, _I, _I, _I, _I, _I, _I, C.POINTER(_I), _I, _I]
_INV = [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I]
_FWD2 = [_P, _P] + _FWD[1:]
_INV2 = [_P, _P] + _INV[1:]

# libdwt entry points (void functions: they log + abort() on failure, like the
# reference).  The Python wrappers below go through dwt_hip_transform2d instead so
# that a failure surfaces as an exception rather than killing the interpreter.
for _n, _sig in (("dwt_cdf97_2f_s", _FWD), ("dwt_cdf97_2i_s", _INV), ("dwt_cdf97_2f_s2", _FWD2),
                 ("dwt_cdf97_2i_s2", _INV2), ("dwt_cdf53_2f_i", _FWD), ("dwt_cdf53_2i_i", _INV),
                 ("dwt_cdf53_2f_s", _FWD), ("dwt_cdf53_2i_s", _INV), ("dwt_cdf97_2f_d", _FWD), ("dwt_cdf97_2i_d", _INV),
                 ("dwt_cdf53_2f_d", _FWD), ("dwt_cdf53_2i_d", _INV), ("dwt_cdf97_2f_i", _FWD), ("dwt_cdf97_2i_i", _INV)):
    getattr(lib, _n).argtypes = _sig
    getattr(lib, _n).restype = None

lib.dwt_hip_init.restype = _I
lib.dwt_hip_device_count.restype = _I
lib.dwt_hip_set_device.argtypes = [_I]
lib.dwt_hip_set_device.restype = _I
lib.dwt_hip_get_device.restype = _I
lib.dwt_hip_device_name.restype = C.c_char_p
lib.dwt_hip_last_error.restype = C.c_char_p
lib.dwt_hip_set_stream.argtypes = [_P]
lib.dwt_hip_set_workspace.argtypes = [_P, C.c_size_t, _P, C.c_size_t]
lib.dwt_hip_set_workspace.restype = _I
lib.dwt_hip_transform2d_batch_sharded.argtypes = [_I, _I, _P, _P, C.c_size_t, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), _I]
lib.dwt_hip_transform2d_batch_sharded.restype = _I
_PP = C.POINTER(_P)
lib.dwt_hip_transform2d_batch_multi.argtypes = [_I, _I, _PP, _PP, C.POINTER(_I), C.POINTER(_I), _I, C.c_size_t, _I, _I, _I, C.POINTER(_I)]
lib.dwt_hip_transform2d_batch_multi.restype = _I
lib.dwt_hip_tune_batch_multi.argtypes = [_I, _I, _PP, _PP, C.POINTER(_I), C.POINTER(_I), _I, C.c_size_t, _I, _I, _I, _I]
lib.dwt_hip_tune_batch_multi.restype = _I
lib.dwt_hip_shard_bounds.argtypes = [_I, _I, _I, C.POINTER(_I), C.POINTER(_I)]
lib.dwt_hip_shard_bounds.restype = None
lib.dwt_hip_tune.argtypes = [_I, _I, _P, _P, C.c_size_t, _I, _I, _I, _I, _I]
lib.dwt_hip_tune.restype = _I
lib.dwt_hip_grant_access.argtypes = [_P, C.POINTER(_I), _I]
lib.dwt_hip_grant_access.restype = _I
lib.dwt_hip_alloc_batch_note.restype = C.c_char_p
lib.dwt_hip_alloc_batch.argtypes = [_I, _I, _I, _I, _I, C.POINTER(_P), C.POINTER(_P)]
lib.dwt_hip_alloc_batch.restype = _I
lib.dwt_hip_placement_report.argtypes = [C.POINTER(C.c_double), _I]
lib.dwt_hip_placement_report.restype = _I
lib.dwt_hip_alloc_volumes.argtypes = [_I, _I, _I, _I, C.POINTER(_P), C.POINTER(_P)]
lib.dwt_hip_alloc_volumes.restype = _I
lib.dwt_hip_alloc_batch_report.argtypes = [C.POINTER(_I)] * 5 + [C.POINTER(C.c_double)] * 2
lib.dwt_hip_probe_pair_us.argtypes = [_P, _P, C.c_size_t]
lib.dwt_hip_probe_pair_us.restype = C.c_double
lib.dwt_hip_probe_copy_us.argtypes = [_P, _P, C.c_size_t]
lib.dwt_hip_probe_copy_us.restype = C.c_double
lib.dwt_hip_malloc_mapped.argtypes = [C.c_size_t, C.c_size_t, _I, C.c_size_t]
lib.dwt_hip_malloc_mapped.restype = _P
lib.dwt_hip_free_mapped.argtypes = [_P]
lib.dwt_hip_malloc_spread.argtypes = [C.c_size_t, C.c_size_t]
lib.dwt_hip_malloc_spread.restype = _P
lib.dwt_hip_set_option.argtypes = [C.c_char_p, _I]
lib.dwt_hip_set_option.restype = _I
lib.dwt_hip_get_option.argtypes = [C.c_char_p]
lib.dwt_hip_get_option.restype = _I
lib.dwt_hip_transform2d.argtypes = [_I, _I, _P, _P, _I, _I, _I, _I, _I, _I, C.POINTER(_I), _I, _I]
lib.dwt_hip_transform2d.restype = _I
lib.dwt_hip_transform2d_batch.argtypes = [_I, _I, _P, _P, _S, _I, _I, _I, _I, C.POINTER(_I)]
lib.dwt_hip_transform2d_batch.restype = _I
lib.dwt_hip_transform2d_interleaved.argtypes = [_I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, C.POINTER(_I), _I]
lib.dwt_hip_transform2d_interleaved.restype = _I
lib.dwt_hip_transform3d.argtypes = [_I, _P, _S, _S, _I, _I, _I, _I]
lib.dwt_hip_transform3d.restype = _I
lib.dwt_hip_transform3d_op.argtypes = [_P, _P, _S, _S, _I, _I, _I, _I]
lib.dwt_hip_transform3d_op.restype = _I
lib.dwt_hip_volume_fwd_op.argtypes = [_P, _S, _S, _P, _S, _S, _I, _I, _I, _I]
lib.dwt_hip_volume_fwd_op.restype = _I
lib.dwt_hip_volume_ip.argtypes = [_I, _P, _S, _S, _I, _I, _I]
lib.dwt_hip_volume_ip.restype = _I
lib.dwt_hip_malloc.argtypes = [_S]
lib.dwt_hip_malloc.restype = _P
lib.dwt_hip_free.argtypes = [_P]
lib.dwt_hip_memcpy_h2d.argtypes = [_P, _P, _S]
lib.dwt_hip_memcpy_h2d.restype = _I
lib.dwt_hip_memcpy_d2h.argtypes = [_P, _P, _S]
lib.dwt_hip_memcpy_d2h.restype = _I
lib.dwt_hip_is_device_pointer.argtypes = [_P]
lib.dwt_hip_is_device_pointer.restype = _I
lib.dwt_hip_prof_enable.argtypes = [_I]
lib.dwt_hip_prof_read.argtypes = [C.POINTER(C.c_double), C.POINTER(_I)]
lib.dwt_hip_prof_read.restype = _I
lib.dwt_hip_prof_read_levels.argtypes = [C.POINTER(C.c_double), C.POINTER(_I), _I]
lib.dwt_hip_prof_read_levels.restype = _I
lib.dwt_util_get_opt_stride.argtypes = [_I]
lib.dwt_util_get_opt_stride.restype = _I
lib.dwt_util_get_stride.argtypes = [_I, _I]
lib.dwt_util_get_stride.restype = _I
lib.dwt_util_set_accel.argtypes = [_I]
lib.dwt_util_get_accel.restype = _I
for _n in ("dwt_util_test_image_fill_s", "dwt_util_test_image_fill_i"):
    getattr(lib, _n).argtypes = [_P, _I, _I, _I, _I, _I]
    getattr(lib, _n).restype = None
for _n in ("dwt_util_compare_s", "dwt_util_compare_i"):
    getattr(lib, _n).argtypes = [_P, _P, _I, _I, _I, _I]
    getattr(lib, _n).restype = _I
for _n in ("dwt_util_conv_show_s", "dwt_util_conv_show_i", "dwt_util_copy_s", "dwt_util_copy_i"):
    getattr(lib, _n).argtypes = [_P, _P, _I, _I, _I, _I]
    getattr(lib, _n).restype = None
lib.dwt_util_save_to_pgm_s.argtypes = [C.c_char_p, C.c_float, _P, _I, _I, _I, _I]
lib.dwt_util_save_to_pgm_s.restype = _I
lib.dwt_util_save_to_pgm_i.argtypes = [C.c_char_p, _I, _P, _I, _I, _I, _I]
lib.dwt_util_save_to_pgm_i.restype = _I
lib.dwt_util_version.restype = C.c_char_p


def _addr(obj):
    """Address of a numpy array / torch tensor / ctypes buffer / int."""
    if obj is None:
        raise DwtError("null image")
    if isinstance(obj, int):
        return obj
    if hasattr(obj, "data_ptr"):  # torch tensor (host or device)
        return obj.data_ptr()
    if hasattr(obj, "ctypes"):  # numpy
        return obj.ctypes.data
    return C.cast(obj, C.c_void_p).value


def last_error():
    return lib.dwt_hip_last_error().decode(errors="replace")


def _check(rc, what):
    if rc:
        raise DwtError(f"{what}: {last_error()}")


# ---- lifecycle -----------------------------------------------------------------------
def dwt_util_init():
    _check(lib.dwt_hip_init(), "dwt_util_init")


def dwt_util_finish():
    lib.dwt_hip_finish()


def dwt_util_set_accel(accel_type):
    lib.dwt_util_set_accel(accel_type)


def dwt_util_get_accel():
    return lib.dwt_util_get_accel()


def device_count():
    return lib.dwt_hip_device_count()


def set_device(device):
    """Bind the calling thread's context to `device` (one host thread per GPU drives several GPUs)."""
    _check(lib.dwt_hip_set_device(int(device)), "dwt_hip_set_device")


def get_device():
    return lib.dwt_hip_get_device()


def device_name():
    return lib.dwt_hip_device_name().decode()


def set_stream(stream_handle):
    """Run subsequent transforms on this hipStream_t (int handle; 0 = default)."""
    lib.dwt_hip_set_stream(stream_handle)


def use_torch_stream():
    import torch

    lib.dwt_hip_set_stream(torch.cuda.current_stream().cuda_stream)


def sync():
    lib.dwt_hip_sync()


def set_option(name, value):
    _check(lib.dwt_hip_set_option(name.encode(), int(value)), "dwt_hip_set_option")


def get_option(name):
    return lib.dwt_hip_get_option(name.encode())


# ---- the reference's 2-D entry points --------------------------------------------------
def _fwd(wavelet, src, dst, stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, zero_padding, who):
    j = _I(j_max)
    rc = lib.dwt_hip_transform2d(wavelet, 0, _addr(src), _addr(dst), stride_x, stride_y, sox, soy, six, siy,
                                 C.byref(j), decompose_one, zero_padding)
    _check(rc, who)
    return j.value


def _inv(wavelet, src, dst, stride_x, stride_y, sox, soy, six, siy, j_max, decompose_one, zero_padding, who):
    j = _I(j_max)
    rc = lib.dwt_hip_transform2d(wavelet, 1, _addr(src), _addr(dst), stride_x, stride_y, sox, soy, six, siy,
                                 C.byref(j), decompose_one, zero_padding)
    _check(rc, who)


def dwt_cdf97_2f_s(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                   j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:12776.  Returns the level count the C function stores in *j_max_ptr."""
    return _fwd(CDF97_S, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                j_max, decompose_one, zero_padding, "dwt_cdf97_2f_s")


def dwt_cdf97_2i_s(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                   j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:17040"""
    _inv(CDF97_S, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
         j_max, decompose_one, zero_padding, "dwt_cdf97_2i_s")


def dwt_cdf97_2f_s2(src, dst, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                    j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:12619"""
    return _fwd(CDF97_S, src, dst, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                j_max, decompose_one, zero_padding, "dwt_cdf97_2f_s2")


def dwt_cdf97_2i_s2(src, dst, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                    j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:17985"""
    _inv(CDF97_S, src, dst, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
         j_max, decompose_one, zero_padding, "dwt_cdf97_2i_s2")


def dwt_cdf53_2f_i(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                   j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:16304"""
    return _fwd(CDF53_I, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                j_max, decompose_one, zero_padding, "dwt_cdf53_2f_i")


def dwt_cdf53_2i_i(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                   j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:18142"""
    _inv(CDF53_I, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
         j_max, decompose_one, zero_padding, "dwt_cdf53_2i_i")


def dwt_cdf53_2f_s(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                   j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:16470"""
    return _fwd(CDF53_S, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                j_max, decompose_one, zero_padding, "dwt_cdf53_2f_s")


def dwt_cdf53_2i_s(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                   j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:18296"""
    _inv(CDF53_S, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
         j_max, decompose_one, zero_padding, "dwt_cdf53_2i_s")


def dwt_cdf97_2f_d(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                   j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:12451 (double precision; stride_y = 8)"""
    return _fwd(CDF97_D, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                j_max, decompose_one, zero_padding, "dwt_cdf97_2f_d")


def dwt_cdf97_2i_d(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                   j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:16884"""
    _inv(CDF97_D, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
         j_max, decompose_one, zero_padding, "dwt_cdf97_2i_d")


def dwt_cdf53_2f_d(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                   j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:12535"""
    return _fwd(CDF53_D, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                j_max, decompose_one, zero_padding, "dwt_cdf53_2f_d")


def dwt_cdf53_2i_d(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                   j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:16962"""
    _inv(CDF53_D, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
         j_max, decompose_one, zero_padding, "dwt_cdf53_2i_d")


def dwt_cdf97_2f_i(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                   j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:16387 (fixed-point int32 CDF 9/7)"""
    return _fwd(CDF97_I, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                j_max, decompose_one, zero_padding, "dwt_cdf97_2f_i")


def dwt_cdf97_2i_i(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                   j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:18219"""
    _inv(CDF97_I, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
         j_max, decompose_one, zero_padding, "dwt_cdf97_2i_i")


FORWARD = {"cdf97_s": dwt_cdf97_2f_s, "cdf53_i": dwt_cdf53_2f_i, "cdf53_s": dwt_cdf53_2f_s,
           "cdf97_d": dwt_cdf97_2f_d, "cdf53_d": dwt_cdf53_2f_d, "cdf97_i": dwt_cdf97_2f_i}
INVERSE = {"cdf97_s": dwt_cdf97_2i_s, "cdf53_i": dwt_cdf53_2i_i, "cdf53_s": dwt_cdf53_2i_s,
           "cdf97_d": dwt_cdf97_2i_d, "cdf53_d": dwt_cdf53_2i_d, "cdf97_i": dwt_cdf97_2i_i}
WAVELET_ID = {"cdf97_s": CDF97_S, "cdf53_i": CDF53_I, "cdf53_s": CDF53_S, "cdf97_d": CDF97_D, "cdf53_d": CDF53_D,
              "cdf97_i": CDF97_I}


# ---- interleaved (in-place lifting) layout ------------------------------------------------
def transform2d_interleaved(wavelet, inverse, flavour, src, dst, stride_x, stride_y, size_o_big_x, size_o_big_y,
                            size_i_big_x=None, size_i_big_y=None, j_max=-1, decompose_one=0):
    """dwt_hip_transform2d_interleaved: host or device pointers, in place (src is dst) or out of place.
    flavour 0 = libdwt.h *_inplace_s entries, 1 = dwt-simple.h fdwt2_* (forward only).  Returns j."""
    j = _I(j_max)
    six = size_o_big_x if size_i_big_x is None else size_i_big_x
    siy = size_o_big_y if size_i_big_y is None else size_i_big_y
    rc = lib.dwt_hip_transform2d_interleaved(WAVELET_ID.get(wavelet, wavelet), int(inverse), flavour, _addr(src), _addr(dst),
                                             stride_x, stride_y, size_o_big_x, size_o_big_y, six, siy, C.byref(j), decompose_one)
    _check(rc, "dwt_hip_transform2d_interleaved")
    return j.value


def dwt_cdf97_2f_inplace_s(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                           j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:12926.  Returns the level count."""
    return transform2d_interleaved(CDF97_S, 0, 0, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y,
                                   size_i_big_x, size_i_big_y, j_max, decompose_one)


def dwt_cdf97_2i_inplace_s(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                           j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:17474"""
    transform2d_interleaved(CDF97_S, 1, 0, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y,
                            size_i_big_x, size_i_big_y, j_max, decompose_one)


def dwt_cdf53_2f_inplace_s(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                           j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:16553.  Returns the level count."""
    return transform2d_interleaved(CDF53_S, 0, 0, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y,
                                   size_i_big_x, size_i_big_y, j_max, decompose_one)


def dwt_cdf53_2i_inplace_s(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                           j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:17886"""
    transform2d_interleaved(CDF53_S, 1, 0, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y,
                            size_i_big_x, size_i_big_y, j_max, decompose_one)


def dwt_cdf97_2f_inplace_i(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                           j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:17424 (fixed-point int 9/7, interleaved).  Returns the level count."""
    return transform2d_interleaved(CDF97_I, 0, 0, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y,
                                   size_i_big_x, size_i_big_y, j_max, decompose_one)


def dwt_cdf97_2i_inplace_i(ptr, stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y,
                           j_max=-1, decompose_one=0, zero_padding=0):
    """src/libdwt.c:17308"""
    transform2d_interleaved(CDF97_I, 1, 0, ptr, ptr, stride_x, stride_y, size_o_big_x, size_o_big_y,
                            size_i_big_x, size_i_big_y, j_max, decompose_one)


def _newapi(wavelet):
    def f(ptr, size_x, size_y, stride_x, stride_y, j_max=-1, decompose_one=0):
        return transform2d_interleaved(wavelet, 0, 1, ptr, ptr, stride_x, stride_y, size_x, size_y, size_x, size_y,
                                       j_max, decompose_one)
    return f


# src/dwt-simple.c:2224 / :1615 / :3034 and :2356 / :1927 / :3166 (argument order of dwt-simple.h)
fdwt2_cdf97_horizontal_s = fdwt2_cdf97_vertical_s = fdwt2_cdf97_diagonal_s = _newapi(CDF97_S)
fdwt2_cdf53_horizontal_s = fdwt2_cdf53_vertical_s = fdwt2_cdf53_diagonal_s = _newapi(CDF53_S)


# ---- batches resident in HBM -----------------------------------------------------------
def transform2d_batch(wavelet, inverse, src, dst, batch_stride, batch, stride_x, size_x, size_y, j_max=-1):
    j = _I(j_max)
    rc = lib.dwt_hip_transform2d_batch(WAVELET_ID.get(wavelet, wavelet), int(inverse), _addr(src), _addr(dst),
                                       batch_stride, batch, stride_x, size_x, size_y, C.byref(j))
    _check(rc, "dwt_hip_transform2d_batch")
    return j.value


def transform2d_batch_sharded(wavelet, inverse, src, dst, batch_stride, batch, stride_x, size_x, size_y, j_max, devices):
    """dwt_hip_transform2d_batch_sharded: the batch split over `devices` (one process, one host thread per slot)."""
    j = _I(j_max)
    dv = (_I * len(devices))(*devices)
    rc = lib.dwt_hip_transform2d_batch_sharded(WAVELET_ID.get(wavelet, wavelet), int(inverse), _addr(src), _addr(dst),
                                               batch_stride, batch, stride_x, size_x, size_y, C.byref(j), dv, len(devices))
    _check(rc, "dwt_hip_transform2d_batch_sharded")
    return j.value


def shard_bounds(batch, n_slots, slot):
    """dwt_hip_shard_bounds: (first image, number of images) of `slot` when image b belongs to slot b*n_slots//batch."""
    a, n = _I(), _I()
    lib.dwt_hip_shard_bounds(batch, n_slots, slot, C.byref(a), C.byref(n))
    return a.value, n.value


def _multi_args(srcs, dsts, counts, devices):
    n = len(srcs)
    assert len(dsts) == n and len(counts) == n and len(devices) == n
    # (`None if p is None`: truth-testing a tensor or array with several elements raises, and a one-element zero
    # tensor must not turn into NULL)
    ptrs = lambda ps: (_P * n)(*[None if p is None else _addr(p) for p in ps])  # noqa: E731
    return ptrs(srcs), ptrs(dsts), (_I * n)(*counts), (_I * n)(*devices), n


def transform2d_batch_multi(wavelet, inverse, srcs, dsts, counts, devices, batch_stride, stride_x, size_x, size_y, j_max=-1):
    """dwt_hip_transform2d_batch_multi: shard k (counts[k] images at srcs[k] / dsts[k]) is RESIDENT on devices[k];
    all shards are transformed at once, nothing crosses xGMI."""
    j = _I(j_max)
    s_, d_, c_, v_, n = _multi_args(srcs, dsts, counts, devices)
    rc = lib.dwt_hip_transform2d_batch_multi(WAVELET_ID.get(wavelet, wavelet), int(inverse), s_, d_, c_, v_, n, batch_stride, stride_x,
                                             size_x, size_y, C.byref(j))
    _check(rc, "dwt_hip_transform2d_batch_multi")
    return j.value


def tune_batch_multi(wavelet, inverse, srcs, dsts, counts, devices, batch_stride, stride_x, size_x, size_y, levels=-1):
    """dwt_hip_tune_batch_multi: dwt_hip_tune in every slot of a resident sharded batch."""
    s_, d_, c_, v_, n = _multi_args(srcs, dsts, counts, devices)
    _check(lib.dwt_hip_tune_batch_multi(WAVELET_ID.get(wavelet, wavelet), int(inverse), s_, d_, c_, v_, n, batch_stride, stride_x,
                                        size_x, size_y, levels), "dwt_hip_tune_batch_multi")


def tune(wavelet, inverse, src, dst, batch_stride, batch, stride_x, size_x, size_y, levels=-1):
    """dwt_hip_tune: the explicit measurement (scratch placement, tile heights) on the caller's own device buffers;
    `dst` receives the transform of `src`.  The calling thread's context keeps the results."""
    _check(lib.dwt_hip_tune(WAVELET_ID.get(wavelet, wavelet), int(inverse), _addr(src), _addr(dst), batch_stride, batch, stride_x,
                            size_x, size_y, levels), "dwt_hip_tune")


def grant_access(ptr, devices):
    dv = (_I * len(devices))(*devices)
    _check(lib.dwt_hip_grant_access(_addr(ptr), dv, len(devices)), "dwt_hip_grant_access")


def alloc_batch_note():
    """'' when the last alloc_batch / alloc_volumes of this thread ran its search, else why it allocated plainly."""
    return lib.dwt_hip_alloc_batch_note().decode()


def transform3d(inverse, vol, stride_y, stride_z, size_x, size_y, size_z, levels=1):
    _check(lib.dwt_hip_transform3d(int(inverse), _addr(vol), stride_y, stride_z, size_x, size_y, size_z, levels),
           "dwt_hip_transform3d")


def transform3d_op(src, dst, stride_y, stride_z, size_x, size_y, size_z, levels=1):
    """Forward 3-D transform out of place (cdf97_3f_op_sep_horizontal_s, src/volume-dwt.c:727)."""
    _check(lib.dwt_hip_transform3d_op(_addr(src), _addr(dst), stride_y, stride_z, size_x, size_y, size_z, levels),
           "dwt_hip_transform3d_op")


# ---- struct volume_t (include/volume.h, include/volume-dwt.h) ---------------------------
class volume_t(C.Structure):
    """The reference's 3-D container (src/volume.h:14-24): sizes, byte strides, data pointer."""
    _fields_ = [("size_x", _I), ("size_y", _I), ("size_z", _I), ("stride_x", _S), ("stride_y", _S), ("stride_z", _S),
                ("data", _P)]


def volume_of(arr_or_ptr, shape_zyx=None, strides_zyx=None):
    """A volume_t over a (z, y, x) float32 numpy array / torch tensor, or over a raw pointer with
    explicit shape and byte strides.  The caller keeps the memory alive."""
    if shape_zyx is None:
        shape_zyx = tuple(arr_or_ptr.shape)
        if hasattr(arr_or_ptr, "data_ptr"):
            strides_zyx = tuple(s * arr_or_ptr.element_size() for s in arr_or_ptr.stride())
        else:
            strides_zyx = tuple(arr_or_ptr.strides)
    nz, ny, nx = shape_zyx
    sz, sy, sx = strides_zyx
    return volume_t(nx, ny, nz, sx, sy, sz, _addr(arr_or_ptr))


_VP = C.POINTER(volume_t)
for _n in ("volume_alloc_realiably", "volume_alloc_realiably_locked", "volume_alloc_device"):
    getattr(lib, _n).argtypes = [_S, _I, _I, _I, _I]
    getattr(lib, _n).restype = _VP
for _n in ("volume_free", "volume_fill_s", "volume_invalidate_cache", "cdf97_3f_ip_sep_horizontal_s", "cdf97_3i_ip_sep_horizontal_s"):
    getattr(lib, _n).argtypes = [_VP]
    getattr(lib, _n).restype = None
for _n in ("volume_copy_s", "volume_compare_s"):
    getattr(lib, _n).argtypes = [_VP, _VP]
    getattr(lib, _n).restype = _I
VOLUME_OP_SCHEDULES = ("sep_horizontal", "sep_vertical", "slices_vert4x4", "baseline_vert2x2x2", "HORIZ_vert2x2x2",
                       "cube_vert4x4x2", "HORIZ_vert4x4x2", "HORIZ_vert4x4x4", "baseline_diag2x2x2", "HORIZ_diag2x2x2")
for _n in VOLUME_OP_SCHEDULES:
    getattr(lib, "cdf97_3f_op_%s_s" % _n).argtypes = [_VP, _VP]
    getattr(lib, "cdf97_3f_op_%s_s" % _n).restype = None
lib.cdf97_3f_op_wrapper_s.argtypes = [_VP, _VP, _I]
lib.cdf97_3f_op_wrapper_s.restype = None
lib.volume_save_to_pgm_s.argtypes = [_VP, C.c_char_p]
lib.volume_save_to_pgm_s.restype = None
lib.volume_perftest_fwd97op_s.argtypes = [_I, _I, _I, _I, C.POINTER(C.c_double), C.POINTER(C.c_ulong)]
lib.volume_perftest_fwd97op_s.restype = _I
lib.volume_perftest_fwd97op_device_s.argtypes = [_I, _I, _I, _I, C.POINTER(C.c_double)]
lib.volume_perftest_fwd97op_device_s.restype = _I
lib.volume_measure_fwd97op_s.argtypes = [_I, _I, _I, _I, _I, _I]
lib.volume_measure_fwd97op_s.restype = _I


def cdf97_3f_ip_sep_horizontal_s(volume):
    """src/volume-dwt.c:677: forward, in place (a volume_t; host or device data)."""
    lib.cdf97_3f_ip_sep_horizontal_s(C.byref(volume))


def cdf97_3i_ip_sep_horizontal_s(volume):
    """src/volume-dwt.c:1115: inverse, in place."""
    lib.cdf97_3i_ip_sep_horizontal_s(C.byref(volume))


def cdf97_3f_op_sep_horizontal_s(volume_src, volume_dst):
    """src/volume-dwt.c:727: forward, out of place."""
    lib.cdf97_3f_op_sep_horizontal_s(C.byref(volume_src), C.byref(volume_dst))


def cdf97_3f_op_wrapper_s(volume_src, volume_dst, approach):
    """src/volume-dwt.c:2787: the schedule dispatcher (enum volume_approach 0..12)."""
    lib.cdf97_3f_op_wrapper_s(C.byref(volume_src), C.byref(volume_dst), int(approach))


def volume_perftest_fwd97op_s(size, opt_stride, approach, N, device=False):
    """src/volume-dwt.c:2810: (errors, seconds per voxel); device=True keeps both volumes in HBM."""
    secs = C.c_double()
    if device:
        err = lib.volume_perftest_fwd97op_device_s(size, opt_stride, int(approach), N, C.byref(secs))
    else:
        faults = C.c_ulong()
        err = lib.volume_perftest_fwd97op_s(size, opt_stride, int(approach), N, C.byref(secs), C.byref(faults))
    return err, secs.value


# ---- device memory without torch -------------------------------------------------------
class DeviceImage:
    """A dense device-resident image (hipMalloc) addressed like libdwt images."""

    def __init__(self, height, width, itemsize=4, pitch_bytes=None):
        self.h, self.w = height, width
        self.stride_x = pitch_bytes or width * itemsize
        self.stride_y = itemsize
        self.nbytes = self.stride_x * height
        self.ptr = lib.dwt_hip_malloc(max(self.nbytes, 16))
        if not self.ptr:
            raise DwtError("dwt_hip_malloc: " + last_error())

    def upload(self, arr):
        import numpy as np

        a = np.ascontiguousarray(arr)
        assert a.nbytes == self.nbytes, (a.nbytes, self.nbytes)
        _check(lib.dwt_hip_memcpy_h2d(self.ptr, a.ctypes.data, self.nbytes), "h2d")
        return self

    def download(self, dtype):
        import numpy as np

        out = np.empty((self.h, self.stride_x // np.dtype(dtype).itemsize), dtype=dtype)
        _check(lib.dwt_hip_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes), "d2h")
        return out

    def free(self):
        if self.ptr:
            lib.dwt_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- kernel timing ----------------------------------------------------------------------
def prof_enable(on=True):
    """True/1: time the level-0 kernel; 2: time every level's kernel."""
    lib.dwt_hip_prof_enable(int(on))


def prof_read_levels(n=8):
    ms = (C.c_double * n)()
    cnt = (_I * n)()
    _check(lib.dwt_hip_prof_read_levels(ms, cnt, n), "dwt_hip_prof_read_levels")
    return [(ms[i] / cnt[i] if cnt[i] else 0.0) for i in range(n)], list(cnt)


def prof_read():
    """(summed ms of the level-0 sweep kernel launches, number of launches) since last read."""
    ms, n = C.c_double(0), _I(0)
    _check(lib.dwt_hip_prof_read(C.byref(ms), C.byref(n)), "dwt_hip_prof_read")
    return ms.value, n.value


def placement_report():
    """Milliseconds the last placement search measured per candidate (empty: no search ran) and the index kept."""
    ms = (C.c_double * 8)()
    n = lib.dwt_hip_placement_report(ms, 8)
    return [round(ms[i], 4) for i in range(n)], lib.dwt_hip_get_option(b"place_last_best")


def alloc_batch_report():
    """What the last dwt_hip_alloc_batch of this thread measured (chunks == 0: plain allocations)."""
    v = [_I() for _ in range(5)]
    ms = (C.c_double * 5)()
    sec = C.c_double()
    lib.dwt_hip_alloc_batch_report(*[C.byref(x) for x in v], ms, C.byref(sec))
    return {"arena_GiB": v[0].value, "dst_positions_tried": v[1].value, "scratch_positions_tried": v[2].value,
            "dst_at_GiB": v[3].value, "scratch_at_GiB": v[4].value,
            "dst_one_level_ms_best_worst": [round(ms[0], 4), round(ms[1], 4)],
            "whole_call_ms_best_worst": [round(ms[2], 4), round(ms[3], 4)], "kept_arrangement_ms": round(ms[4], 4), "seconds": round(sec.value, 2)}


def alloc_batch(wavelet, n_images, size_x, size_y, levels=-1):
    """dwt_hip_alloc_batch: (src, dst) device pointers of a resident batch, placed; free with lib.dwt_hip_free."""
    s_, d_ = _P(), _P()
    _check(lib.dwt_hip_alloc_batch(WAVELET_ID.get(wavelet, wavelet), n_images, size_x, size_y, levels, C.byref(s_), C.byref(d_)), "dwt_hip_alloc_batch")
    return s_.value, d_.value


def alloc_volumes(size_x, size_y, size_z, levels):
    """dwt_hip_alloc_volumes: (src, dst) dense device volumes of an out-of-place 3-D call, placed; free with lib.dwt_hip_free."""
    s_, d_ = _P(), _P()
    _check(lib.dwt_hip_alloc_volumes(size_x, size_y, size_z, levels, C.byref(s_), C.byref(d_)), "dwt_hip_alloc_volumes")
    return s_.value, d_.value
