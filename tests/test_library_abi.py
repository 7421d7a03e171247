"""CPU-only checks of the shipped shared library: it loads, exports every symbol the
public headers declare, and its host-side helpers behave like the reference's.  No
compute entry is called here (there is no GPU in the build container)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")


@pytest.fixture(scope="module")
def dwt():
    import __graft_entry__ as g

    if not os.path.exists(os.path.join(ROOT, "libdwt_amd", "libdwt_hip.so")):
        g.build()
    import libdwt_amd

    return libdwt_amd


def declared_functions(header):
    text = open(os.path.join(INCLUDE, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"enum\s+\w+\s*\{.*?\};", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:dwt|fdwt2)_\w+)\s*\(", text)))


@pytest.mark.parametrize("header", ["libdwt.h", "libdwt_hip.h", "dwt-simple.h"])
def test_exports_every_declared_symbol(dwt, header):
    names = declared_functions(header)
    assert len(names) > (15 if header != "dwt-simple.h" else 5)
    missing = [n for n in names if not hasattr(dwt.lib, n)]
    assert not missing, missing


def test_headers_compile_as_c99_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "libdwt.h"\n#include "libdwt_hip.h"\n#include "dwt-simple.h"\nint main(void){int j=-1;(void)j;return 0;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", INCLUDE, "-c", str(src), "-o", str(tmp_path / "t.o")])
    subprocess.check_call(["g++", "-x", "c++", "-std=c++11", "-Wall", "-Werror", "-I", INCLUDE, "-c", str(src), "-o", str(tmp_path / "t2.o")])


def test_product_never_references_the_oracle():
    """The product path must not route through oracle/ (tier rule 3)."""
    pkg = os.path.join(ROOT, "libdwt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oraclelib" not in text and "dwt_oracle" not in text and "libdwt_ref" not in text, f
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libdwt_hip.so")]).decode()
    assert "oracle" not in out and "libdwt_ref" not in out


def test_transform_fails_loudly_without_gpu(dwt):
    if dwt.device_count() > 0:
        pytest.skip("a GPU is present")
    img = np.zeros((16, 16), np.float32)
    with pytest.raises(dwt.DwtError) as e:
        dwt.dwt_cdf97_2f_s(img, 64, 4, 16, 16, 16, 16, -1)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)
    assert not img.any()  # nothing was computed behind our back


def test_strides_match_reference(dwt, reference):
    for n in [1, 2, 3, 64, 2048, 2052, 32768, 4 * 1000, 340, 341 * 4, 561 * 4 + 1]:
        assert dwt.lib.dwt_util_get_opt_stride(n) == reference.lib.dwt_util_get_opt_stride(n), n
        for opt in range(9):
            assert dwt.lib.dwt_util_get_stride(n, opt) == reference.lib.dwt_util_get_stride(n, opt), (n, opt)
    assert dwt.lib.dwt_util_get_opt_stride(2048) == 2053  # examples/simple: 512 floats


def test_test_images_and_compare(dwt, oracle):
    a = np.zeros((37, 41), np.float32)
    b = np.zeros((37, 41), np.float32)
    dwt.lib.dwt_util_test_image_fill_s(a.ctypes.data, a.strides[0], 4, 41, 37, 0)
    oracle.fill_s(b)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    ai = np.zeros((37, 41), np.int32)
    bi = np.zeros((37, 41), np.int32)
    dwt.lib.dwt_util_test_image_fill_i(ai.ctypes.data, ai.strides[0], 4, 41, 37, 1)
    oracle.fill_i(bi, 1)
    assert np.array_equal(ai, bi)
    # compare: float eps 1e-3 absolute, NaN/Inf => differ; int exact
    c = a.copy()
    assert dwt.lib.dwt_util_compare_s(a.ctypes.data, c.ctypes.data, a.strides[0], 4, 41, 37) == 0
    c[3, 4] += 5e-4
    assert dwt.lib.dwt_util_compare_s(a.ctypes.data, c.ctypes.data, a.strides[0], 4, 41, 37) == 0
    c[3, 4] += 1e-2
    assert dwt.lib.dwt_util_compare_s(a.ctypes.data, c.ctypes.data, a.strides[0], 4, 41, 37) == 1
    c = a.copy()
    c[0, 0] = np.nan
    assert dwt.lib.dwt_util_compare_s(c.ctypes.data, c.ctypes.data, a.strides[0], 4, 41, 37) == 1
    ci = ai.copy()
    assert dwt.lib.dwt_util_compare_i(ai.ctypes.data, ci.ctypes.data, ai.strides[0], 4, 41, 37) == 0
    ci[36, 40] += 1
    assert dwt.lib.dwt_util_compare_i(ai.ctypes.data, ci.ctypes.data, ai.strides[0], 4, 41, 37) == 1


def test_conv_show_copy_pgm(dwt, tmp_path):
    a = (np.arange(12, dtype=np.float32).reshape(3, 4) - 5) / 4
    out = np.zeros_like(a)
    dwt.lib.dwt_util_conv_show_s(a.ctypes.data, out.ctypes.data, 16, 4, 4, 3)
    want = (np.log(1.0 + np.abs(a).astype(np.float32) * np.float32(100.0)).astype(np.float32) / np.float32(10)).astype(np.float32)
    assert np.allclose(out, want, rtol=1e-6)
    i = np.array([[-3, 4], [0, -7]], np.int32)
    oi = np.zeros_like(i)
    dwt.lib.dwt_util_conv_show_i(i.ctypes.data, oi.ctypes.data, 8, 4, 2, 2)
    assert np.array_equal(oi, np.abs(i))
    cp = np.zeros_like(a)
    dwt.lib.dwt_util_copy_s(a.ctypes.data, cp.ctypes.data, 16, 4, 4, 3)
    assert np.array_equal(cp, a)
    p = tmp_path / "x.pgm"
    img = np.array([[0.0, 0.5], [1.0, 0.25]], np.float32)
    assert dwt.lib.dwt_util_save_to_pgm_s(str(p).encode(), 1.0, img.ctypes.data, 8, 4, 2, 2) == 0
    toks = p.read_text().split()
    assert toks[:4] == ["P2", "2", "2", "255"] and [int(t) for t in toks[4:]] == [0, 127, 255, 63]


def test_reference_examples_link_unchanged(dwt, tmp_path):
    """The reference's own programs around the path -- examples/simple, simple-int,
    simple-perf, subbands, perf-plot, simple-newapi -- compile against include/*.h and link against
    libdwt_hip.so without modification."""
    ref = "/root/reference/examples"
    if not os.path.isdir(ref):
        pytest.skip("reference sources not present on this machine")
    libdir = os.path.join(ROOT, "libdwt_amd")
    import glob

    for ex in ("simple", "simple-int", "simple-perf", "subbands", "perf-plot", "simple-newapi"):
        exe = tmp_path / (ex + ".bin")
        src = sorted(glob.glob(os.path.join(ref, ex, "*.c")))[0]
        subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", INCLUDE, src,
                               "-o", str(exe), "-L", libdir, "-l:libdwt_hip.so", "-Wl,-rpath," + libdir, "-lm"])
        assert exe.exists()
