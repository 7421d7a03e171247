cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "device_resident or config2 or golden_host" > gpurun_out/r02/t16.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02/t16.log
python scripts/single_levels.py "inplace_defer=1" "inplace_defer=0" "inplace_defer=1" "inplace_defer=0" > gpurun_out/r02/single_levels5.log 2>&1; cat gpurun_out/r02/single_levels5.log
for w in config3 config4 config5; do python bench.py --workload $w --steps 20 --warmup 5 2>/dev/null; done > gpurun_out/r02/bench_other_r02.jsonl; cat gpurun_out/r02/bench_other_r02.jsonl | cut -c1-400
