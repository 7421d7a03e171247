cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_hip_volume.py tests/test_hip_parity.py -x -q -m gpu -k "volume or out_of_place or golden_single or double_precision_batch" > gpurun_out/r02/t21.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r02/t21.log
VARIANTS="vol_rows=8;vol_rows=8;vol_rows=8" timeout -k 10 300 python scripts/vol_op_bench.py 1024 3 > gpurun_out/r02/vol_bench5.log 2>&1; cat gpurun_out/r02/vol_bench5.log
