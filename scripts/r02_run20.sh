cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_hip_volume.py tests/test_hip_parity.py -x -q -m gpu -k "out_of_place_forward or double_precision_batch" > gpurun_out/r02/t20.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r02/t20.log
VARIANTS="vol_dpp=0;vol_dpp=1;vol_dpp=0;vol_dpp=1" timeout -k 10 300 python scripts/vol_op_bench.py 1024 3 > gpurun_out/r02/vol_bench4.log 2>&1; cat gpurun_out/r02/vol_bench4.log
