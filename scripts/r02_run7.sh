cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 500 python -m pytest tests/test_hip_interleaved.py -x -q -m gpu > gpurun_out/r02/t7.log 2>&1; echo "tests rc=$?"; tail -30 gpurun_out/r02/t7.log
