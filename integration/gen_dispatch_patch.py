#!/usr/bin/env python3
"""Writes integration/libdwt_hip_dispatch.patch: the lines a libdwt maintainer ADDS to route the hot path to the
MI355X backend, as a unified diff with zero context lines against libdwt 2015-02-18-dev.

Every hunk is a pure insertion (`@@ -N,0 +M,k @@` followed by `+` lines only): the patch carries no line of the
reference, only positions in it.  The positions are found by reading the reference's sources where they lie
(the opening brace of each driver, the include block, the 3-D dispatcher); run in the build container:

    python integration/gen_dispatch_patch.py [/root/reference]

oracle/Makefile `ref_hybrid` pipes `patch -o -` into the compiler: no patched source is ever written to disk."""
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

ARGS10 = "stride_x, stride_y, size_o_big_x, size_o_big_y, size_i_big_x, size_i_big_y"
# (function, wavelet id, inverse, source expr, destination expr)
DRIVERS = [
    ("dwt_cdf97_2f_s2", "DWT_HIP_CDF97_S", 0, "src", "dst"),
    ("dwt_cdf97_2f_s", "DWT_HIP_CDF97_S", 0, "ptr", "ptr"),
    ("dwt_cdf53_2f_i", "DWT_HIP_CDF53_I", 0, "ptr", "ptr"),
    ("dwt_cdf53_2f_s", "DWT_HIP_CDF53_S", 0, "ptr", "ptr"),
    ("dwt_cdf97_2i_s", "DWT_HIP_CDF97_S", 1, "ptr", "ptr"),
    ("dwt_cdf97_2i_s2", "DWT_HIP_CDF97_S", 1, "src", "dst"),
    ("dwt_cdf53_2i_i", "DWT_HIP_CDF53_I", 1, "ptr", "ptr"),
    ("dwt_cdf53_2i_s", "DWT_HIP_CDF53_S", 1, "ptr", "ptr"),
]


def body_open(lines, name):
    """1-based number of the line holding the opening brace of `void name(`'s body."""
    for i, ln in enumerate(lines):
        if re.match(r"void %s\((\)|void\))?$" % re.escape(name), ln.rstrip()):
            for k in range(i, i + 40):
                if lines[k].rstrip() == "{":
                    return k + 1
    raise SystemExit(f"{name}: not found")


def line_of(lines, pattern, first=0):
    for i in range(first, len(lines)):
        if re.match(pattern, lines[i].rstrip()):
            return i + 1
    raise SystemExit(f"{pattern}: not found")


def driver_hunk(name, wav, inverse, s, d):
    j = "&j_hip" if inverse else "j_max_ptr"
    out = ["#ifdef WITH_HIP_BACKEND",
           "\tif( DWT_ACCEL_HIP == get_accel_type() )",
           "\t{"]
    if inverse:
        out.append("\t\tint j_hip = j_max;")
    out += [f"\t\tif( dwt_hip_transform2d({wav}, {inverse}, {s}, {d}, {ARGS10}, {j}, decompose_one, zero_padding) )",
            "\t\t\tdwt_util_error(\"%s: %s\\n\", __func__, dwt_hip_last_error());",
            "\t\treturn;",
            "\t}",
            "#endif"]
    return out


def emit(path_in_ref, inserts):
    """inserts: [(after_line, [new lines])] -> unified diff text with zero context"""
    out = [f"--- a/{path_in_ref}", f"+++ b/{path_in_ref}"]
    shift = 0
    for after, new in sorted(inserts):
        out.append(f"@@ -{after},0 +{after + shift + 1},{len(new)} @@")
        out += ["+" + ln for ln in new]
        shift += len(new)
    return out


def main():
    src = open(os.path.join(REF, "src/libdwt.c"), errors="replace").read().split("\n")
    ins = []
    # the backend's C-ABI and the new accel value, after the system includes
    ins.append((line_of(src, r'#include <ctype\.h>'), [
        "#ifdef WITH_HIP_BACKEND",
        "#include \"libdwt_hip.h\" /* the MI355X backend's C-ABI: -I<repo>/include, link -l:libdwt_hip.so */",
        "#define DWT_ACCEL_HIP 100 /* dwt_util_set_accel(100): the 2-D drivers run on the GPU */",
        "#endif"]))
    for name, wav, inverse, s, d in DRIVERS:
        ins.append((body_open(src, name), driver_hunk(name, wav, inverse, s, d)))
    ins.append((body_open(src, "dwt_util_init"), [
        "#ifdef WITH_HIP_BACKEND",
        "\tif( dwt_hip_init() )",
        "\t\tdwt_util_log(LOG_WARN, \"%s: %s (accel %i is not available)\\n\", __func__, dwt_hip_last_error(), DWT_ACCEL_HIP);",
        "\telse if( getenv(\"LIBDWT_ACCEL\") ) /* programs that never call dwt_util_set_accel: LIBDWT_ACCEL=100 ./simple */",
        "\t\tdwt_util_set_accel(atoi(getenv(\"LIBDWT_ACCEL\")));",
        "#endif"]))
    ins.append((body_open(src, "dwt_util_finish"), [
        "#ifdef WITH_HIP_BACKEND",
        "\tdwt_hip_finish();",
        "#endif"]))
    text = emit("src/libdwt.c", ins)

    vol = open(os.path.join(REF, "src/volume-dwt.c"), errors="replace").read().split("\n")
    vins = []
    typedef = line_of(vol, r"typedef void \(\*volume_func_t\)")
    vins.append((typedef - 1, [
        "#ifdef WITH_HIP_BACKEND",
        "#include \"libdwt_hip.h\"",
        "#define VOL_HIP 100 /* cdf97_3f_op_wrapper_s(src, dst, VOL_HIP): x, y and z lifting in ONE fused pass on the GPU */",
        "void cdf97_3f_op_hip_s(struct volume_t *volume_src, struct volume_t *volume_dst)",
        "{",
        "\t/* host or device `data`, each volume with its own strides; 7 = x, y and z */",
        "\tif( dwt_hip_volume_fwd_op(volume_src->data, volume_src->stride_y, volume_src->stride_z, volume_dst->data, volume_dst->stride_y, volume_dst->stride_z,",
        "\t\tvolume_dst->size_x, volume_dst->size_y, volume_dst->size_z, 7) )",
        "\t\tdwt_util_error(\"%s: %s\\n\", __func__, dwt_hip_last_error());",
        "}",
        "void cdf97_3i_ip_hip_s(struct volume_t *volume)",
        "{",
        "\tif( dwt_hip_volume_ip(1, volume->data, volume->stride_y, volume->stride_z, volume->size_x, volume->size_y, volume->size_z) )",
        "\t\tdwt_util_error(\"%s: %s\\n\", __func__, dwt_hip_last_error());",
        "}",
        "#endif"]))
    wrapper = line_of(vol, r"void cdf97_3f_op_wrapper_s\(", typedef)
    brace = next(k + 1 for k in range(wrapper - 1, wrapper + 5) if vol[k].rstrip() == "{")
    vins.append((brace, [
        "#ifdef WITH_HIP_BACKEND",
        "\tif( VOL_HIP == (int)approach )",
        "\t{",
        "\t\tcdf97_3f_op_hip_s(volume_src, volume_dst);",
        "\t\treturn;",
        "\t}",
        "#endif"]))
    text += emit("src/volume-dwt.c", vins)
    head = ["libdwt (2015-02-18-dev) -> MI355X backend: the WITH_HIP_BACKEND dispatch of INTEGRATION.md s2.",
            "Insert-only hunks (zero context): apply with `patch -p1` in a libdwt tree of that version, build with",
            "-DWITH_HIP_BACKEND -I<repo>/include and link -L<repo>/libdwt_amd -l:libdwt_hip.so.",
            "Generated by integration/gen_dispatch_patch.py; built and tested by `make -C oracle ref_hybrid` +",
            "tests/test_hip_hybrid.py.", ""]
    with open(os.path.join(HERE, "libdwt_hip_dispatch.patch"), "w") as f:
        f.write("\n".join(head + text) + "\n")
    print("wrote", os.path.join(HERE, "libdwt_hip_dispatch.patch"), len(text), "lines")


if __name__ == "__main__":
    main()
