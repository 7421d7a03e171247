"""GPU: INTEGRATION.md s2 made real.  `make -C oracle ref_hybrid` compiles the reference itself with the maintainer's
dispatch lines (integration/libdwt_hip_dispatch.patch: insert-only hunks at the top of the eight 2-D drivers, in
dwt_util_init / dwt_util_finish and in the 3-D schedule dispatcher) and links it against libdwt_hip.so.  In that ONE
binary `dwt_util_set_accel(0)` is libdwt's own CPU code and `dwt_util_set_accel(100)` the MI355X backend: the two
must agree bit for bit, entry by entry; and the reference's own programs linked against the hybrid run both ways."""
import ctypes as C
import filecmp
import os
import subprocess

import numpy as np
import pytest

from conftest import bits, full_range_ints
from oraclelib import HYBRID_DIR, HYBRID_SO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hyb():
    from oraclelib import Hybrid

    if not os.path.exists(HYBRID_SO) and not os.path.isdir("/root/reference/src"):
        pytest.skip("oracle/_ref/hybrid was not built (needs the reference's sources at build time)")
    h = Hybrid()
    h.lib.dwt_util_init()
    yield h
    h.lib.dwt_util_set_accel(0)
    h.lib.dwt_util_finish()


ENTRIES = [("cdf97_2f_s", "cdf97_2i_s", np.float32), ("cdf53_2f_i", "cdf53_2i_i", np.int32), ("cdf53_2f_s", "cdf53_2i_s", np.float32)]


@pytest.mark.parametrize("ff,fi,dt", ENTRIES, ids=[e[0] for e in ENTRIES])
def test_accel_100_equals_accel_0_in_one_binary(hyb, ff, fi, dt):
    rng = np.random.default_rng(100)
    for (h, w, j, d1) in [(512, 512, -1, 0), (300, 1030, 3, 0), (37, 53, -1, 1), (1024, 2048, 5, 0)]:
        a = full_range_ints(rng, (h, w)) if dt == np.int32 else rng.random((h, w), dtype=np.float32) * 2 - 1
        cpu, gpu = a.copy(), a.copy()
        hyb.lib.dwt_util_set_accel(0)
        jc = hyb.fwd(ff, cpu, j, decompose_one=d1)
        hyb.lib.dwt_util_set_accel(100)
        jg = hyb.fwd(ff, gpu, j, decompose_one=d1)
        assert jc == jg and np.array_equal(bits(cpu), bits(gpu)), (ff, h, w)
        hyb.lib.dwt_util_set_accel(0)
        hyb.inv(fi, cpu, jc, decompose_one=d1)
        hyb.lib.dwt_util_set_accel(100)
        hyb.inv(fi, gpu, jg, decompose_one=d1)
        assert np.array_equal(bits(cpu), bits(gpu)), (fi, h, w)
    # sparse frame with zero padding and the reference's prime pitch
    a = rng.random((64, 64), dtype=np.float32) if dt != np.int32 else rng.integers(-999, 999, (64, 64), dtype=np.int32)
    cpu, gpu = a.copy(), a.copy()
    kw = dict(size_o=(64, 64), size_i=(50, 40), zero_padding=1)
    hyb.lib.dwt_util_set_accel(0)
    hyb.fwd(ff, cpu, -1, **kw)
    hyb.lib.dwt_util_set_accel(100)
    hyb.fwd(ff, gpu, -1, **kw)
    hyb.lib.dwt_util_set_accel(0)
    assert np.array_equal(bits(cpu), bits(gpu))


def test_s2_entries_accel_100_equals_accel_0(hyb):
    rng = np.random.default_rng(101)
    src = rng.random((700, 900), dtype=np.float32)
    outs = []
    for accel in (0, 100):
        hyb.lib.dwt_util_set_accel(accel)
        dst = np.full_like(src, 7.0)
        j = hyb.call2("cdf97_2f_s2", src.copy(), dst, 4)
        rec = np.full_like(src, 3.0)
        hyb.call2("cdf97_2i_s2", dst, rec, j)
        outs.append((j, dst, rec))
    hyb.lib.dwt_util_set_accel(0)
    assert outs[0][0] == outs[1][0] and np.array_equal(bits(outs[0][1]), bits(outs[1][1])) and np.array_equal(bits(outs[0][2]), bits(outs[1][2]))


def test_vol_hip_schedule_equals_sep_horizontal(hyb):
    """cdf97_3f_op_wrapper_s(src, dst, VOL_HIP = 100) against VOL_SEP_HORIZONTAL = 0 of the same binary, and the
    bound inverse cdf97_3i_ip_hip_s against cdf97_3i_ip_sep_horizontal_s."""

    class Vol(C.Structure):
        _fields_ = [("size_x", C.c_int), ("size_y", C.c_int), ("size_z", C.c_int), ("stride_x", C.c_size_t),
                    ("stride_y", C.c_size_t), ("stride_z", C.c_size_t), ("data", C.c_void_p)]

    def vol(a):
        return Vol(a.shape[2], a.shape[1], a.shape[0], a.strides[2], a.strides[1], a.strides[0], a.ctypes.data)

    rng = np.random.default_rng(102)
    for shp in [(24, 40, 256), (9, 7, 6), (33, 35, 300)]:
        v = rng.random(shp, dtype=np.float32)
        cpu, gpu = np.zeros_like(v), np.zeros_like(v)
        hyb.lib.cdf97_3f_op_wrapper_s(C.byref(vol(v)), C.byref(vol(cpu)), 0)
        hyb.lib.cdf97_3f_op_wrapper_s(C.byref(vol(v)), C.byref(vol(gpu)), 100)
        assert np.array_equal(bits(cpu), bits(gpu)), shp
        hyb.lib.cdf97_3i_ip_sep_horizontal_s(C.byref(vol(cpu)))
        hyb.lib.cdf97_3i_ip_hip_s(C.byref(vol(gpu)))
        assert np.array_equal(bits(cpu), bits(gpu)), shp


def test_reference_programs_linked_against_the_hybrid(tmp_path):
    """examples/simple of the reference, linked against the hybrid: run as it is (libdwt's CPU path) and with
    LIBDWT_ACCEL=100 (the hook dwt_util_init gained: the GPU) -- same verdict, byte-identical data*.pgm;
    examples/test (which loops over accel 0 .. 16 itself) still reports success 34 times."""
    exe = os.path.join(HYBRID_DIR, "simple")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/hybrid was not built")
    runs = {}
    for tag, accel in (("cpu", None), ("gpu", "100")):
        d = tmp_path / tag
        d.mkdir()
        env = dict(os.environ)
        env.pop("LIBDWT_ACCEL", None)
        if accel:
            env["LIBDWT_ACCEL"] = accel
        env.setdefault("OMP_NUM_THREADS", "4")
        out = subprocess.run([exe], cwd=d, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-1500:]
        assert "success" in (out.stdout + out.stderr).lower()
        runs[tag] = d
    files = sorted(os.listdir(runs["cpu"]))
    assert files and files == sorted(os.listdir(runs["gpu"]))
    for f in files:
        assert filecmp.cmp(runs["cpu"] / f, runs["gpu"] / f, shallow=False), f
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    out = subprocess.run([os.path.join(HYBRID_DIR, "test")], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and (out.stdout + out.stderr).lower().count("success") >= 34, (out.stdout + out.stderr)[-1500:]
