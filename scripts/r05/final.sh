#!/bin/bash
# Round-5 artefacts in one call: the four bench lines, the entries' wall clock and trace, the headline's kernel trace +
# counters, the counter pass over every entry's kernels.  The counter passes come LAST (they leave the box's PCIe in a state
# that slows host-pointer calls until the next fresh box).
#   gpurun --timeout 1200 -- 'bash scripts/r05/final.sh'   (then: python scripts/r05/collect.py)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
python bench.py > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.err; echo "headline rc=$?"
: > gpurun_out/r05/bench_other_configs.jsonl
for w in config3 config4 config5; do
  python bench.py --workload $w >> gpurun_out/r05/bench_other_configs.jsonl 2>> gpurun_out/r05/bench.err; echo "$w rc=$?"
done
DWT_HIP_TUNE=1 timeout -k 10 300 python scripts/measure_entries.py > gpurun_out/r05/entries_unprofiled.txt 2>&1; echo "entries rc=$?"
rm -rf gpurun_out/prof_entries_r05
DWT_HIP_TUNE=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_entries_r05 -- python3 scripts/measure_entries.py > gpurun_out/r05/entries_profiled.log 2>&1; echo "entries trace rc=$?"
bash scripts/profile_gpu.sh r05 > gpurun_out/r05/profile.log 2>&1; echo "profile rc=$?"
cp gpurun_out/r05/bench.json gpurun_out/bench_r05.json  # (summarize_profile.py quotes the un-profiled line of the same call)
python3 scripts/summarize_profile.py gpurun_out/prof_r05 r05 > gpurun_out/r05/summarize.log 2>&1; echo "summarize rc=$?"
python3 scripts/summarize_entries.py gpurun_out/prof_entries_r05 gpurun_out/r05/entries_profiled.log r05 > gpurun_out/r05/summarize_entries.log 2>&1; echo "summarize entries rc=$?"
cp profiles/r05_kernel_stats.csv profiles/r05_pmc_level0.json profiles/r05_summary.md profiles/r05_entries_kernel_stats.csv profiles/r05_entries_summary.md gpurun_out/r05/ 2>/dev/null
bash scripts/r05/pmc_kernels.sh > gpurun_out/r05/pmc_kernels.log 2>&1; echo "pmc kernels rc=$?"
find gpurun_out/prof_r05 gpurun_out/prof_entries_r05 -name "*.csv" -size +8M -delete
tail -c 300 gpurun_out/r05/bench.json
