#!/bin/bash
# Round-end artefacts: tests, the bench line, its profile + counters, the entries trace, other configs.
#   gpurun --timeout 1200 -- 'bash scripts/archive/r03/r03_final.sh'
set -u
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03/final_gpu_tests.log 2>&1; rc=$?
tail -2 gpurun_out/r03/final_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r03.json 2> gpurun_out/r03/bench_err.log || exit 1
tail -c 600 gpurun_out/bench_r03.json; echo
bash scripts/profile_gpu.sh r03 > gpurun_out/r03/profile.log 2>&1 || exit 1
rm -rf gpurun_out/prof_entries_r03
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_entries_r03 -- python3 scripts/measure_entries.py > gpurun_out/entries_r03.log 2>&1 || exit 1
tail -5 gpurun_out/entries_r03.log
: > gpurun_out/bench_other_r03.jsonl
for w in config3 config4 config5; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu 2>/dev/null | grep '^{' >> gpurun_out/bench_other_r03.jsonl
done
wc -l gpurun_out/bench_other_r03.jsonl
find gpurun_out/prof_r03 gpurun_out/prof_entries_r03 -name "*.csv" -size +8M -delete
