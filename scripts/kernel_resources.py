#!/usr/bin/env python3
"""Register / occupancy table of every kernel of one .hip file, from the compiler's remarks:
    scripts/kernel_resources.py libdwt_amd/csrc/dwt_vol3d_ip.hip [name filter regex]"""
import re
import subprocess
import sys

f = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "."
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off",
       "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-c", f, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, []
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark: (?:\S+ )?\s*([A-Za-z ]+?(?: \[[^\]]+\])?): (\d+)", ln)
    if m and cur is not None and "remark" in ln:
        cur[m.group(1).strip()] = m.group(2)
    elif "error" in ln or "warning:" in ln:
        print(ln, file=sys.stderr)
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    if not re.search(pat, n):
        continue
    g = r.get
    print(f"{n:72s} vgpr {g('VGPRs')} agpr {g('AGPRs')} sgpr {g('SGPRs')} spill s{g('SGPRs Spill')}/v{g('VGPRs Spill')} "
          f"scratch {g('ScratchSize [bytes/lane]')} occ {g('Occupancy [waves/SIMD]')}")
