"""GPU: the reference's OWN example and self-test programs, compiled unchanged from the
reference's sources (oracle/Makefile ref_examples, in the build container) and linked
against the product library, run on the MI355X.  They are the reference's acceptance
test of the drop-in: `examples/test` must print "success" for every accel value and
data type, the simple-* programs must report that the inverse restored the image."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXDIR = os.path.join(ROOT, "oracle", "_ref", "examples")


def expected():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "example_outputs.json")) as f:
        return json.load(f)["examples"]


def write_input_pgm(path, w=300, h=200):
    """The same input image oracle/gen_example_outputs.py feeds to examples/load*."""
    with open(path, "w") as f:
        f.write("P2\n# synthetic input for examples/load\n%d %d\n255\n" % (w, h))
        for y in range(h):
            f.write(" ".join(str(255 * (2 * x * y) // (x * x + y * y + 1)) for x in range(w)) + "\n")


def run(name, tmp_path, timeout=300):
    exe = os.path.join(EXDIR, name)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/examples not built (needs /root/reference at build time)")
    argv = [exe]
    if name in ("load", "load-int"):
        write_input_pgm(os.path.join(str(tmp_path), "input.pgm"))
        argv.append("input.pgm")
    out = subprocess.run(argv, cwd=str(tmp_path), capture_output=True, text=True, timeout=timeout)
    text = out.stdout + out.stderr
    assert out.returncode == 0, text[-2000:]
    return text


def test_reference_self_test_program(tmp_path):
    """examples/test/test.c: 16 accel values x (in-place, out-of-place) float 9/7, then double
    and fixed-point int 9/7: 34 checks."""
    text = run("test", tmp_path)
    assert text.count("success") == 34, text[-3000:]
    assert "fail" not in text


@pytest.mark.parametrize("name", ["simple", "simple-int", "simple-double", "simple-newapi", "simple-single-loop", "subbands",
                                  "subbands-int", "start", "load", "load-int"])
def test_reference_example_programs_write_the_reference_bytes(tmp_path, name):
    """Same verdict lines and byte-identical PGM files as the program produced on the reference
    (tests/golden/example_outputs.json, recorded by oracle/gen_example_outputs.py) -- including
    examples/simple-single-loop, the 9/7 interleaved in-place pair, whose coefficients follow the
    reference's phase order bit for bit."""
    import hashlib
    import re

    want = expected()[name]
    text = re.sub(r"\x1b\[[0-9;]*m", "", run(name, tmp_path))
    verdicts = [l.split("INFO: ")[-1] for l in text.splitlines() if "success" in l or "differs" in l]
    assert verdicts == want["verdicts"], text[-2000:]
    got = {f: hashlib.sha256(open(os.path.join(str(tmp_path), f), "rb").read()).hexdigest() for f in want["files"]}
    assert got == want["files"]


@pytest.mark.parametrize("name", ["simple-perf", "simple-perf-int", "simple-perf-line", "simple-perf-single", "simple-perf-single-sdl"])
def test_reference_perf_programs(tmp_path, name):
    text = run(name, tmp_path, timeout=600)
    assert "rror" not in text and "differs" not in text, text[-2000:]


def test_reference_perf_plot_program(tmp_path):
    """examples/perf-plot: libdwt's size sweep (dwt_util_measure_perf_cdf97_2_s over array types,
    strides, accel values) writes its gnuplot data files."""
    os.makedirs(os.path.join(str(tmp_path), "data"))
    text = run("perf-plot", tmp_path, timeout=600)
    assert "rror" not in text, text[-2000:]
    files = os.listdir(os.path.join(str(tmp_path), "data"))
    assert len(files) >= 10
    first = open(os.path.join(str(tmp_path), "data", sorted(files)[0])).read().split()
    assert len(first) >= 2 and float(first[1]) > 0
