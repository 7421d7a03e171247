cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "cdf97_d or cdf53_d" > gpurun_out/r02/t18.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r02/t18.log
python scripts/d_bench.py 8192 5 > gpurun_out/r02/d_bench3.log 2>&1; cat gpurun_out/r02/d_bench3.log
