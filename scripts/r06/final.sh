#!/bin/bash
# Round-6 artefacts in one call: the four bench lines, the entries' wall clock and trace, the headline's kernel trace +
# counters, the counter pass over every entry's kernels.  The counter passes come LAST (they leave the box's PCIe in a state
# that slows host-pointer calls until the next fresh box).
#   gpurun --timeout 1200 -- 'bash scripts/r06/final.sh'   (then: python scripts/r06/collect.py)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
python bench.py > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err; echo "headline rc=$?"
: > gpurun_out/r06/bench_other_configs.jsonl
for w in config3 config4 config5; do
  python bench.py --workload $w >> gpurun_out/r06/bench_other_configs.jsonl 2>> gpurun_out/r06/bench.err; echo "$w rc=$?"
done
DWT_HIP_TUNE=1 timeout -k 10 300 python scripts/measure_entries.py > gpurun_out/r06/entries_unprofiled.txt 2>&1; echo "entries rc=$?"
rm -rf gpurun_out/prof_entries_r06
DWT_HIP_TUNE=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_entries_r06 -- python3 scripts/measure_entries.py > gpurun_out/r06/entries_profiled.log 2>&1; echo "entries trace rc=$?"
python scripts/strided_device_timing.py > gpurun_out/r06/strided_device.json 2>> gpurun_out/r06/bench.err; echo "strided rc=$?"
rm -rf gpurun_out/prof_strided_r06
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_strided_r06 -- python3 scripts/strided_device_timing.py > gpurun_out/r06/strided_profiled.log 2>&1; echo "strided trace rc=$?"
find gpurun_out/prof_strided_r06 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06/strided_kernel_stats.csv \;
bash scripts/profile_gpu.sh r06 > gpurun_out/r06/profile.log 2>&1; echo "profile rc=$?"
cp gpurun_out/r06/bench.json gpurun_out/bench_r06.json  # (summarize_profile.py quotes the un-profiled line of the same call)
python3 scripts/summarize_profile.py gpurun_out/prof_r06 r06 > gpurun_out/r06/summarize.log 2>&1; echo "summarize rc=$?"
python3 scripts/summarize_entries.py gpurun_out/prof_entries_r06 gpurun_out/r06/entries_profiled.log r06 > gpurun_out/r06/summarize_entries.log 2>&1; echo "summarize entries rc=$?"
cp profiles/r06_kernel_stats.csv profiles/r06_pmc_level0.json profiles/r06_summary.md profiles/r06_entries_kernel_stats.csv profiles/r06_entries_summary.md gpurun_out/r06/ 2>/dev/null
bash scripts/r06/pmc_kernels.sh > gpurun_out/r06/pmc_kernels.log 2>&1; echo "pmc kernels rc=$?"
find gpurun_out/prof_r06 gpurun_out/prof_entries_r06 -name "*.csv" -size +8M -delete
tail -c 300 gpurun_out/r06/bench.json
