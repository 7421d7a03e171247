cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_hip_volume.py -x -q -m gpu > gpurun_out/r02/t13.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02/t13.log
VARIANTS="vol_rows=8;vol_rows=8" timeout -k 10 300 python scripts/vol_op_bench.py 1024 3 > gpurun_out/r02/vol_bench3.log 2>&1; cat gpurun_out/r02/vol_bench3.log
python bench.py --workload config5 --steps 20 --warmup 5 > gpurun_out/r02/bench_config5.json 2>/dev/null; cat gpurun_out/r02/bench_config5.json
