cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02/t19.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r02/t19.log
python scripts/d_bench.py 8192 5 > gpurun_out/r02/d_bench4.log 2>&1; cat gpurun_out/r02/d_bench4.log
