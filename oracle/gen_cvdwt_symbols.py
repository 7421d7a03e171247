#!/usr/bin/env python3
"""List the libdwt functions the reference's OpenCV wrapper calls (src/cvdwt.cpp) into
tests/golden/cvdwt_symbols.json.  TEST INFRASTRUCTURE ONLY; run where /root/reference exists.

OpenCV is not in this image, so the wrapper itself cannot be compiled here; what can be pinned
is (a) that every libdwt entry it calls is exported by the product library
(tests/test_library_abi.py) and (b) that those entries accept its calling convention --
interleaved channels, stride_y = elemSize -- which tests/golden/multichannel.npz covers."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/cvdwt.cpp"


def called_functions(text):
    text = re.sub(r"//[^\n]*", "", text)            # comments mention names that are not calls
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(dwt_[a-z0-9_]+)\s*\(", text))
    # functions the wrapper defines itself (static helpers) are not library symbols
    defined = set(re.findall(r"\bvoid\s+(dwt_[a-z0-9_]+)\s*\(", text))
    return sorted(names - defined)


if __name__ == "__main__":
    syms = called_functions(open(SRC).read())
    out = os.path.join(ROOT, "tests", "golden", "cvdwt_symbols.json")
    with open(out, "w") as f:
        json.dump({"source": "src/cvdwt.cpp (libdwt 2015-02-18-dev)", "generator": "oracle/gen_cvdwt_symbols.py",
                   "calls": syms}, f, indent=1)
    print(out, len(syms), syms)
