#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts/profile_gpu.sh into profiles/<tag>_*.
    python scripts/summarize_profile.py gpurun_out/prof_r01 r01
Writes profiles/<tag>_kernel_stats.csv (the --stats table, torch's own kernels dropped),
profiles/<tag>_pmc_level0.json (per-launch counters of the dominant kernel with the
gfx950 FETCH_SIZE correction applied) and a short profiles/<tag>_summary.md."""
import csv, glob, json, os, sys, collections

def newest(pattern):
    """the most recent match (a profile directory may hold the files of earlier runs)"""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]


src, tag = sys.argv[1], sys.argv[2]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(out, exist_ok=True)

stats = list(csv.DictReader(open(newest(f"{src}/trace/*/*_kernel_stats.csv")[0])))
keep = [r for r in stats if "dwt::" in r["Name"]]
with open(f"{out}/{tag}_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
    w.writeheader()
    w.writerows(keep)

trace = list(csv.DictReader(open(newest(f"{src}/trace/*/*_kernel_trace.csv")[0])))
sweeps = [r for r in trace if "k_fwd_sweep" in r["Kernel_Name"]]
gmax = max(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) for r in sweeps)
l0_rows = [r for r in sweeps if int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) == gmax]
l0_rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the TIMED steps are the last launches of the run (before them: first-placement run, the placement search of
# dwt_hip_alloc_batch -- hundreds of launches on other arrangements -- and the warm-up)
STEPS = int(os.environ.get("STEPS", 5))
l0_rows = l0_rows[-STEPS:]
l0 = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in l0_rows]
l0_name = l0_rows[-1]["Kernel_Name"].split("(")[0].replace("void ", "")
per_level = collections.OrderedDict()
sweeps.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in sweeps[-5 * STEPS:]:  # the timed steps' five levels
    key = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]), r["Kernel_Name"].split("(")[0])
    per_level.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

pmc = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    fs = newest(f"{src}/{d}/*/*_counter_collection.csv")
    if not fs:
        continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "k_fwd_sweep" in r["Kernel_Name"]]
    if not rows:
        continue
    g = max(int(r["Grid_Size"]) for r in rows)
    acc = collections.defaultdict(list)
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        if int(r["Grid_Size"]) == g:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[-STEPS:]  # the timed steps (see above)
        pmc[k] = sum(v) / len(v)

# images per launch = grid.y of the level-0 launch (one blockIdx.y per image); env overrides
gy = max(int(r["Grid_Size_Y"]) for r in sweeps if int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) == gmax)
images = int(os.environ.get("IMAGES_PER_LAUNCH", gy))
summary = {
    "tag": tag,
    "kernel": f"{l0_name} (level 0: {images} images of 8192x8192 float per launch; the template that ran, from the trace)",
    "level0_avg_ns": sum(l0) / len(l0), "level0_launches": len(l0),
    "algorithmic_bytes_per_launch": 2 * 4 * 8192 * 8192 * images,
    "pmc_per_launch": pmc,
}
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE is in KiB and on gfx950 reports exactly half
    # of a wide (16 B/lane) streaming read -> double it; WRITE_SIZE (KiB) is exact.
    fetch = pmc["FETCH_SIZE"] * 1024 * 2
    write = pmc["WRITE_SIZE"] * 1024
    summary["hbm_traffic_bytes_per_launch"] = fetch + write
    summary["hbm_fetch_bytes_corrected"] = fetch
    summary["hbm_write_bytes"] = write
    summary["traffic_over_algorithmic"] = (fetch + write) / summary["algorithmic_bytes_per_launch"]
if "TCC_HIT_sum" in pmc:
    summary["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
json.dump(summary, open(f"{out}/{tag}_pmc_level0.json", "w"), indent=1)

with open(f"{out}/{tag}_summary.md", "w") as f:
    f.write(f"# rocprofv3 summary {tag}\n\nCommand: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-single --no-sweep` "
            f"(+ separate `--pmc` passes), MI355X, see scripts/profile_gpu.sh.\n\n## per-kernel (kernel-trace --stats)\n\n")
    f.write("| kernel | calls | avg us | total % |\n|---|---|---|---|\n")
    for r in keep:
        f.write(f"| `{r['Name'].split('(')[0]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |\n")
    f.write("\n## per level (grid size, kernel) -> avg duration us\n\n")
    for (g, name), v in per_level.items():
        f.write(f"- grid {g} `{name}`: {sum(v)/len(v)/1e3:.1f} us over {len(v)} launches\n")
    f.write(f"\n## dominant kernel (level 0, {summary['level0_launches']} timed launches)\n\n")
    f.write(f"- average duration {summary['level0_avg_ns']/1e3:.1f} us -> algorithmic {summary['algorithmic_bytes_per_launch']/summary['level0_avg_ns']:.0f} GB/s\n")
    if "hbm_traffic_bytes_per_launch" in summary:
        f.write(f"- HBM traffic per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE): {summary['hbm_traffic_bytes_per_launch']/1e6:.1f} MB "
                f"= {summary['traffic_over_algorithmic']:.3f} x algorithmic bytes\n")
    if "l2_hit_rate" in summary:
        f.write(f"- L2 hit rate {summary['l2_hit_rate']:.3f}\n")
    for k in sorted(pmc):
        f.write(f"- {k}: {pmc[k]:.0f}\n")
# agreement between bench.py's HIP-event timing and the profiler (same box, same call)
try:
    bj = None
    for cand in (f"gpurun_out/bench_{tag}.json", f"{out}/{tag}_bench.json"):
        if os.path.exists(cand):
            bj = json.load(open(cand)); break
    tl = [l for l in open(f"{src}/trace.log") if '"metric"' in l]
    uj = json.loads(tl[0][tl[0].index("{"):]) if tl else None
    with open(f"{out}/{tag}_summary.md", "a") as f:
        f.write("\n## agreement between bench.py's HIP-event timing and the profiler\n\n")
        if bj:
            f.write(f"- `bench.py` un-profiled, same GPU box and gpurun call: level-0 launch {bj['roofline']['avg_launch_ms']*1e3:.1f} us by HIP "
                    f"events on the launch stream ({bj['roofline']['achieved']:.0f} GB/s), {bj['value']:.1f} Gsamples/s (`profiles/{tag}_bench.json`).\n")
        if uj:
            f.write(f"- the same command under `rocprofv3 --kernel-trace --stats`: profiler average {summary['level0_avg_ns']/1e3:.1f} us; "
                    f"bench.py's own HIP events inside that profiled run {uj['roofline']['avg_launch_ms']*1e3:.1f} us: the two methods agree "
                    f"under identical conditions.\n")
        f.write("- profiled runs are slower than un-profiled ones (rocprofv3 serialises every dispatch behind a completion signal and the chip "
                "holds lower clocks while profiled: MI355X_MICROARCH.md, DVFS give-back item 2); compare profiled numbers only with profiled numbers.\n")
except Exception as e:
    print("agreement section skipped:", e)
print(json.dumps(summary, indent=1))
