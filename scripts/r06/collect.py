#!/usr/bin/env python3
"""After `gpurun -- bash scripts/r06/final.sh`: copy the summaries the judge reads from gpurun_out/r06/ into profiles/."""
import os, shutil
ROOT = next(d for d in (os.path.abspath(__file__).rsplit(os.sep, k)[0] for k in range(1, 6)) if os.path.exists(os.path.join(d, "bench.py")))
src, dst = os.path.join(ROOT, "gpurun_out", "r06"), os.path.join(ROOT, "profiles")
pairs = {"bench.json": "r06_bench.json", "bench_other_configs.jsonl": "r06_bench_other_configs.jsonl", "entries_unprofiled.txt": "r06_entries_wallclock_unprofiled.txt",
         "r06_kernel_stats.csv": "r06_kernel_stats.csv", "r06_pmc_level0.json": "r06_pmc_level0.json", "r06_summary.md": "r06_summary.md",
         "r06_entries_kernel_stats.csv": "r06_entries_kernel_stats.csv", "r06_entries_summary.md": "r06_entries_summary.md",
         "kernels_pmc.md": "r06_kernels_pmc.md", "strided_device.json": "r06_strided_device.json", "strided_kernel_stats.csv": "r06_strided_kernel_stats.csv"}
for a, b in pairs.items():
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, b)); print("copied", b)
    else:
        print("MISSING", a)
