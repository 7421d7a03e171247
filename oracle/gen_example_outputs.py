#!/usr/bin/env python3
"""Record what the reference's own example programs write when they run on the reference.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):

    make -C oracle ref && python oracle/gen_example_outputs.py

Each program of examples/{simple,simple-int,simple-double,simple-newapi,subbands} is compiled
unchanged against the reference's header and oracle/_ref/libdwt_ref.so, run in a scratch
directory, and the sha256 of every PGM file it writes plus its verdict lines ("success" /
"images differs") go to tests/golden/example_outputs.json.  On the GPU box the same programs,
linked against the product library (oracle/Makefile ref_examples), must write the same bytes
(tests/test_hip_reference_programs.py)."""
import glob
import hashlib
import json
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
EXAMPLES = ["simple", "simple-int", "simple-double", "simple-newapi", "subbands", "subbands-int", "start", "load", "load-int",
            "simple-single-loop"]
# programs that take an input file: a deterministic ASCII PGM written here and, identically, by the GPU test
NEEDS_INPUT = {"load", "load-int"}


def write_input_pgm(path, w=300, h=200):
    """The input image of examples/load*: libdwt's integer test pattern scaled to 0..255."""
    with open(path, "w") as f:
        f.write("P2\n# synthetic input for examples/load\n%d %d\n255\n" % (w, h))
        for y in range(h):
            f.write(" ".join(str(255 * (2 * x * y) // (x * x + y * y + 1)) for x in range(w)) + "\n")



def main():
    refdir = os.path.join(ROOT, "oracle", "_ref")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for ex in EXAMPLES:
            src = sorted(glob.glob(os.path.join(REF, "examples", ex, "*.c")))[0]
            exe = os.path.join(tmp, "ref_" + ex)
            subprocess.check_call(["gcc", "-std=c99", "-O2", "-w", "-fopenmp", "-I", os.path.join(REF, "src"), src, "-o", exe,
                                   "-L", refdir, "-l:libdwt_ref.so", "-Wl,-rpath," + refdir, "-lm"])
            work = os.path.join(tmp, "run_" + ex)
            os.makedirs(work)
            argv = [exe]
            if ex in NEEDS_INPUT:
                write_input_pgm(os.path.join(work, "input.pgm"))
                argv.append("input.pgm")
            text = subprocess.run(argv, cwd=work, capture_output=True, text=True, check=True)
            text = re.sub(r"\x1b\[[0-9;]*m", "", text.stdout + text.stderr)
            files = {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest()
                     for f in sorted(glob.glob(os.path.join(work, "*.pgm"))) if os.path.basename(f) != "input.pgm"}
            verdicts = [l.split("INFO: ")[-1] for l in text.splitlines() if "success" in l or "differs" in l]
            out[ex] = {"files": files, "verdicts": verdicts}
            print(ex, verdicts, len(files), "files")
    with open(os.path.join(ROOT, "tests", "golden", "example_outputs.json"), "w") as f:
        json.dump({"generator": "oracle/gen_example_outputs.py", "reference": "libdwt 2015-02-18-dev", "examples": out}, f, indent=1)


if __name__ == "__main__":
    main()
