cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r02/t8.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r02/t8.log
python scripts/il_bench.py > gpurun_out/r02/il_bench.log 2>&1; tail -12 gpurun_out/r02/il_bench.log
