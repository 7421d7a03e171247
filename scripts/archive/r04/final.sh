#!/bin/bash
# Round-4 artefacts in one call: the entries' wall clock, the four bench lines, the entries trace, the headline's kernel
# trace + counters.  The counter passes come LAST: they leave the box in a state in which PCIe transfers both ways at once
# collapse (the pipelined host-pointer call: 7.4 -> 14.6 ms) until the next fresh box.
#   gpurun --timeout 1200 -- 'bash scripts/archive/r04/final.sh'   (then copy from gpurun_out/ into profiles/)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python scripts/measure_entries.py > gpurun_out/r04_entries_unprofiled.txt 2>&1; echo "entries rc=$?"
bash scripts/archive/r04/bench_all.sh
rm -rf gpurun_out/prof_entries_r04
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_entries_r04 -- python3 scripts/measure_entries.py > gpurun_out/entries_r04.log 2>&1; echo "entries trace rc=$?"
bash scripts/profile_gpu.sh r04 > gpurun_out/r04_profile.log 2>&1; echo "profile rc=$?"
python3 scripts/summarize_profile.py gpurun_out/prof_r04 r04 > gpurun_out/r04_summarize.log 2>&1; echo "summarize rc=$?"
find gpurun_out/prof_r04 gpurun_out/prof_entries_r04 -name "*.csv" -size +8M -delete
cp profiles/r04_kernel_stats.csv profiles/r04_pmc_level0.json profiles/r04_summary.md gpurun_out/ 2>/dev/null  # (written there by summarize_profile.py)
tail -c 400 gpurun_out/r04_bench.json
