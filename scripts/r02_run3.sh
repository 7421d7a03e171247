cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "two_level" > gpurun_out/r02/t3.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r02/t3.log
