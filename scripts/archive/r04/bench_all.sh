#!/bin/bash
# every bench line of round 4 (headline + the other BASELINE configs), one process each
mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; echo "headline rc=$?"
: > gpurun_out/r04_bench_other_configs.jsonl
for w in config3 config4 config5; do
  python bench.py --workload $w >> gpurun_out/r04_bench_other_configs.jsonl 2>> gpurun_out/r04_bench.err; echo "$w rc=$?"
done
