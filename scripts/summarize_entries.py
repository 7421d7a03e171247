#!/usr/bin/env python3
"""profiles/<tag>_entries_{kernel_stats.csv,summary.md} from a rocprofv3 kernel-trace run of
scripts/measure_entries.py:  python scripts/summarize_entries.py gpurun_out/prof_entries gpurun_out/prof_entries.log r01"""
import csv, glob, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src_dir, log, tag = sys.argv[1], sys.argv[2], sys.argv[3]
stats = sorted(glob.glob(os.path.join(src_dir, "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
shutil.copy(stats, os.path.join(ROOT, "profiles", f"{tag}_entries_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
def short(n):
    try:
        d = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n.replace(".kd", "")]).decode().strip()
    except Exception:
        d = n
    d = re.sub(r"\(.*", "", d).replace("void ", "")
    return "torch fill (test data)" if d.startswith("at::native") else d
out = [f"# Round {tag[1:].lstrip('0') or '0'} — every entry of the path under `rocprofv3 --kernel-trace --stats`", "",
       "Command (one MI355X, `scripts/measure_entries.py` runs each entry 10+3 times):", "",
       "    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_entries -- python3 scripts/measure_entries.py", "",
       f"Raw per-kernel table: `profiles/{tag}_entries_kernel_stats.csv`.  Template arguments: "
       "`k_fwd_sweep<wavelet, columns per lane, ring rows, cache policy, interleaved>`, `k_vol_z<inverse, columns per lane, cache policy>`, "
       "`k_vol_fwd_fused<cache policy>`; interleaved layout with the border strips in the launch: `k_fwd_sweep_x<wavelet, columns per lane, ring rows, cache policy>`, `k_inv_sweep_x<wavelet, ring rows, cache policy, split even rows>`.", "",
       "| kernel | calls | avg µs | min µs | max µs | share |", "|---|---|---|---|---|---|"]
for r in rows:
    out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {r['Percentage']} % |")
out += ["", "Wall-clock per entry in the same (profiled) process (`profiles/" + tag + "_entries_wallclock_unprofiled.txt` has the plain run):", "", "```"] + [l.rstrip() for l in open(log) if " us " in l] + ["```", ""]
notes = os.path.join(ROOT, "profiles", f"{tag}_entries_notes.md")  # hand-written additions kept across regenerations
if os.path.exists(notes):
    out += [open(notes).read().rstrip(), ""]
open(os.path.join(ROOT, "profiles", f"{tag}_entries_summary.md"), "w").write("\n".join(out))
print("\n".join(out[:30]))
