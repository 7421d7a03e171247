#!/usr/bin/env python3
"""profiles/r04_shard8_summary.md from a kernel trace of `bench.py --images 8 ...` (the shard of one rank at N = 8):
   python scripts/archive/r04/shard8_summary.py gpurun_out/prof_shard8 gpurun_out/prof_shard8.log"""
import csv, glob, json, os, sys
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 7)) if os.path.exists(os.path.join(d, "bench.py")))
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_fwd_sweep" in r["Kernel_Name"]]
last = rows[-50:]  # the 10 timed steps x 5 levels
out = ["# Round 4 -- the 8-image step (the shard of one rank at N = 8) under rocprofv3 --kernel-trace", "",
       "Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --images 8 --steps 10 --warmup 3 --no-cpu --no-single --no-sweep`", "",
       "The last 10 steps (the timed region), per level: kernel, grid, average duration, average gap to the previous kernel of the step.", "",
       "| level | kernel | grid | avg us | avg gap before us |", "|---|---|---|---|---|"]
for lv in range(5):
    ks = [last[i] for i in range(lv, 50, 5)]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ks]
    gaps = [(int(last[i]["Start_Timestamp"]) - int(last[i - 1]["End_Timestamp"])) / 1e3 for i in range(lv, 50, 5) if i % 5]
    name = ks[0]["Kernel_Name"].split("(")[0].replace("void ", "")
    gx, gy = ks[0]["Grid_Size_X"], ks[0]["Grid_Size_Y"]
    out.append(f"| {lv} | `{name}` | {gx} x {gy} | {sum(dur)/len(dur):.1f} | {(sum(gaps)/len(gaps) if gaps else 0):.1f} |")
steps = [(int(last[i + 4]["End_Timestamp"]) - int(last[i]["Start_Timestamp"])) / 1e3 for i in range(0, 50, 5)]
avg = sum(steps) / len(steps)
out += ["", f"Step (first kernel's start to last kernel's end), average of 10: {avg:.1f} us = {8 * 8192 * 8192 / avg / 1e3:.1f} Gsamples/s per GPU."]
line = [l for l in open(sys.argv[2]) if l.startswith("{")]
if line:
    d = json.loads(line[-1])
    out.append(f"`bench.py` in the same (profiled) run: value {d['value']} Gsamples/s, ms_per_step {d['ms_per_step']}, level 0 {d['roofline']['achieved']} GB/s.")
open(os.path.join(ROOT, "profiles", "r04_shard8_summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
