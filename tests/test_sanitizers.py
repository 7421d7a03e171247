"""CPU-only: the HOST side under sanitizers (GPU AddressSanitizer does not exist on this pool; the reference ships a
valgrind target for the same purpose, common.mk:37-39).

* `tests/san/Makefile asan`: the product's host C files (dwt_entry.c, dwt_util.c, dwt_harness.c, dwt_io.c,
  dwt_volume.c) with a test double in the device backend's place (san_backend_stub.c: "device" memory is host memory, a
  transform is the oracle's), `-fsanitize=address,undefined`, any report aborts.  `san_driver util` walks every
  host utility, the self-test / perf harness over the three frame kinds and the volume helpers.
* The file readers on a corpus of MALFORMED PGM / MAT files: no sanitizer report, and the reference's verdict
  (its return code, or the image it returns) on the same file.
* The oracle itself instrumented, with the whole CPU oracle suite run on it (libasan / libubsan preloaded into python).
* `tests/san/Makefile tsan`: the library's host thread pools under ThreadSanitizer with host-only jobs."""
import ctypes as C
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SAN = os.path.join(HERE, "san")
DRIVER = os.path.join(SAN, "_build", "san_driver")
BAD = ("ERROR: AddressSanitizer", "runtime error:", "ERROR: LeakSanitizer", "WARNING: ThreadSanitizer")


def _gcc_lib(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.fixture(scope="module")
def asan_build():
    if not _gcc_lib("libasan.so") or not _gcc_lib("libubsan.so"):
        pytest.skip("gcc has no sanitizer runtimes here")
    subprocess.check_call(["make", "-s", "-C", SAN, "asan"])
    return DRIVER


def run_clean(cmd, **kw):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.update(kw.pop("env", {}))
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, **kw)
    for b in BAD:
        assert b not in out.stderr and b not in out.stdout, out.stderr[-3000:]
    return out


def test_host_utilities_harness_and_volume_helpers_under_asan_ubsan(asan_build, tmp_path):
    out = run_clean([asan_build, "util", str(tmp_path)])
    assert out.returncode == 0 and "san_driver util OK" in out.stdout, out.stderr[-2000:]


# ---- malformed files -------------------------------------------------------------------------------------------
def corpus(d):
    """name -> (kind, bytes).  kind: pgm (both element types) or mat."""
    big = b"9" * 5000
    files = {
        "pgm_empty": b"",
        "pgm_magic_p5": b"P5\n4 4\n255\n" + bytes(16),
        "pgm_magic_only": b"P2",
        "pgm_comment_only": b"P2\n# nothing but a comment",
        "pgm_comment_eof_in_header": b"P2\n4 # width then EOF",
        "pgm_truncated_header": b"P2\n4\n",
        "pgm_no_depth": b"P2\n4 4\n",
        "pgm_negative_width": b"P2\n-4 4\n255\n" + b"1 " * 16,
        "pgm_zero_size": b"P2\n0 0\n255\n",
        "pgm_zero_width": b"P2\n0 5\n255\n",
        "pgm_negative_maxval": b"P2\n2 2\n-255\n1 2 3 4\n",
        "pgm_zero_maxval": b"P2\n2 2\n0\n0 0 0 0\n",
        "pgm_maxval_mismatch": b"P2\n2 2\n65535\n1 2 3 4\n",
        "pgm_nonnumeric_width": b"P2\nfour 4\n255\n",
        "pgm_nonnumeric_sample": b"P2\n2 2\n255\n1 two 3 4\n",
        "pgm_truncated_data": b"P2\n4 4\n255\n1 2 3 4 5\n",
        "pgm_sample_above_maxval": b"P2\n2 2\n255\n1 2 300 4\n",
        "pgm_negative_sample": b"P2\n2 2\n255\n1 -2 3 4\n",
        "pgm_overlong_token": b"P2\n" + big + b" 4\n255\n",
        "pgm_overlong_sample": b"P2\n2 2\n255\n1 " + big + b" 3 4\n",
        "pgm_int_overflow_dims": b"P2\n2147483647 2147483647\n255\n",
        # 4 * (2^30 + 1) wraps to 4: the reference allocates 5 bytes per row and writes the samples past them
        "pgm_width_wraps_the_pitch": b"P2\n1073741825 1\n255\n" + b"7 " * 4096,
        "pgm_crlf": b"P2\r\n2 2\r\n255\r\n1 2\r\n3 4\r\n",
        "pgm_binary_junk": b"P2\n\x00\xff\xfe\x01 4\n255\n",
        "pgm_extra_data": b"P2\n2 2\n255\n1 2 3 4 5 6 7 8\n",
        "pgm_valid_with_comments": b"P2\n# c1\n3 2 # c2\n255\n# c3\n1 2 3\n4 5 6\n",
        "mat_empty": b"",
        "mat_single": b"3.5\n",
        "mat_ragged": b"1,2,3\n4,5\n6\n",
        "mat_ragged_longer": b"1,2\n3,4,5,6\n",
        "mat_trailing_comma": b"1,2,3,\n4,5,6,\n",
        "mat_leading_comma": b",1,2\n,3,4\n",
        "mat_nonnumeric": b"1,two,3\n4,5,6\n",
        "mat_nan_inf": b"nan,inf,-inf\n1,2,3\n",
        "mat_overlong_token": b"1," + big + b",3\n4,5,6\n",
        "mat_no_newline_at_end": b"1,2\n3,4",
        "mat_blank_lines": b"1,2\n\n3,4\n\n",
        "mat_spaces": b" 1 , 2 \n 3 , 4 \n",
        "mat_semicolons": b"1;2\n3;4\n",
        "mat_exponents": b"1e3,-2.5E-2\n+7,.5\n",
        "mat_huge_row": b",".join([b"1"] * 20000) + b"\n",
        "mat_binary_junk": b"\x00\x01\x02,\xff\n",
        "mat_valid": b"1.5,2.5,3.5\n4.5,5.5,6.5\n",
    }
    out = {}
    for name, data in files.items():
        p = os.path.join(d, name + (".pgm" if name.startswith("pgm") else ".mat"))
        with open(p, "wb") as f:
            f.write(data)
        out[name] = p
    return out


REF_SCRIPT = r"""
import ctypes as C, sys
lib = C.CDLL(sys.argv[1]); kind, path = sys.argv[2], sys.argv[3].encode()
p = C.c_void_p(); sx = C.c_int(); sy = C.c_int(); w = C.c_int(); h = C.c_int()
if kind == "pgm_s":
    lib.dwt_util_load_from_pgm_s.argtypes = [C.c_char_p, C.c_float] + [C.c_void_p] * 5
    rc = lib.dwt_util_load_from_pgm_s(path, 1.0, C.byref(p), C.byref(sx), C.byref(sy), C.byref(w), C.byref(h))
elif kind == "pgm_i":
    lib.dwt_util_load_from_pgm_i.argtypes = [C.c_char_p, C.c_int] + [C.c_void_p] * 5
    rc = lib.dwt_util_load_from_pgm_i(path, 255, C.byref(p), C.byref(sx), C.byref(sy), C.byref(w), C.byref(h))
else:
    fn = getattr(lib, "dwt_util_load_from_" + kind)
    fn.argtypes = [C.c_char_p] + [C.c_void_p] * 5
    rc = fn(path, C.byref(p), C.byref(w), C.byref(h), C.byref(sx), C.byref(sy))
if rc == 0 and p.value:
    t = C.c_int if kind.endswith("_i") else C.c_float
    s = 0.0
    for y in range(h.value):
        for x in range(w.value):
            s += t.from_address(p.value + y * sx.value + x * sy.value).value
    print("rc=0 size=%dx%d sum=%.6g" % (w.value, h.value, s))
else:
    print("rc=%d" % rc)
"""


def verdict(out):
    """What a loader did with a file: its printed verdict, or how the process ended."""
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("rc=")]
    if line:
        return line[-1]
    return "signal %d" % -out.returncode if out.returncode < 0 else "exit %d" % out.returncode


def test_malformed_pgm_and_mat_files(asan_build, reference, tmp_path):
    """Every file of the corpus through the product's loaders (instrumented) and through the reference's: no sanitizer
    report; the same verdict -- the reference's return code, or the same image (size and sum).  Where the REFERENCE itself
    dies on a file (a signal), the product must come back with an error code instead."""
    from oraclelib import REF_SO

    files = corpus(str(tmp_path))
    rows, died = [], []
    for name, path in sorted(files.items()):
        kinds = ("pgm_s", "pgm_i") if name.startswith("pgm") else ("mat_s", "mat_i")
        for kind in kinds:
            ours = run_clean([asan_build, "load", kind, path])
            ref = subprocess.run([sys.executable, "-c", REF_SCRIPT, REF_SO, kind, path], capture_output=True, text=True, timeout=120)
            vo, vr = verdict(ours), verdict(ref)
            rows.append((name, kind, vo, vr))
            if name == "pgm_width_wraps_the_pitch":
                # undefined behaviour in the reference (heap overflow: it may die or seem to succeed); refused here
                assert vo == "rc=2", (name, kind, vo, vr)
                died.append((name, kind, vr, vo))
            elif vr.startswith("signal") or vr.startswith("exit"):
                died.append((name, kind, vr, vo))
                assert vo.startswith("rc=") and vo != "rc=0" or vo.startswith("exit 134"), (name, kind, vo, vr)  # an error code (or the library's own abort)
            else:
                assert vo == vr, (name, kind, "ours: " + vo, "reference: " + vr, ours.stderr[-800:])
    # the corpus must actually exercise both outcomes
    assert any(r[2].startswith("rc=0") for r in rows) and any(r[2] != "rc=0" and r[2].startswith("rc=") for r in rows)
    print("reference died on:", died)


# ---- the oracle instrumented, under the CPU suite ----------------------------------------------------------------
def test_oracle_suite_on_the_instrumented_oracle(asan_build):
    """tests/test_oracle.py, test_oracle_interleaved.py and test_float_range.py with DWT_ORACLE_SO pointing at the
    ASan + UBSan build of the oracle and the two runtimes preloaded into the interpreter."""
    so = os.path.join(SAN, "_build", "libdwt_oracle_san.so")
    env = {"LD_PRELOAD": _gcc_lib("libasan.so") + ":" + _gcc_lib("libubsan.so"), "DWT_ORACLE_SO": so,
           "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0"}
    out = run_clean([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                     os.path.join(HERE, "test_oracle.py"), os.path.join(HERE, "test_oracle_interleaved.py"),
                     os.path.join(HERE, "test_float_range.py")], env=env, cwd=os.path.dirname(HERE))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert " passed" in out.stdout


def test_host_thread_pools_under_tsan():
    """libdwt_amd/csrc/dwt_host_pools.h (the RowPool of the host-pointer calls, the SlotThread workers of the multi-GPU
    entries) compiled for the host with -fsanitize=thread around host-only jobs: three caller threads taking turns on
    the row pool, 200 submit / wait rounds on four slot workers with failing jobs among them."""
    if not _gcc_lib("libtsan.so"):
        pytest.skip("gcc has no ThreadSanitizer runtime here")
    subprocess.check_call(["make", "-s", "-C", SAN, "tsan"])
    out = subprocess.run([os.path.join(SAN, "_build", "tsan_pools")], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0 and "tsan_pools OK" in out.stdout and "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
