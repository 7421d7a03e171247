cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 500 python -m pytest tests/test_hip_interleaved.py tests/test_hip_reference_programs.py -x -q -m gpu > gpurun_out/r02/t9.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r02/t9.log
python scripts/il_bench.py > gpurun_out/r02/il_bench.log 2>&1; tail -5 gpurun_out/r02/il_bench.log
