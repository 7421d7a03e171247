cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_hip_volume.py -x -q -m gpu > gpurun_out/r02/t12.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02/t12.log
VARIANTS="vol_rows=8;vol_rows=6;vol_rows=6,vol_tile_pairs=64;vol_rows=6,vol_tile_pairs=256;vol_rows=6,vol_nt=7;vol_rows=8;vol_rows=6" timeout -k 10 300 python scripts/vol_op_bench.py 1024 3 > gpurun_out/r02/vol_bench2.log 2>&1; cat gpurun_out/r02/vol_bench2.log
VARIANTS="vol_rows=8;vol_rows=6" timeout -k 10 300 python scripts/vol_op_bench.py 512 3 >> gpurun_out/r02/vol_bench2.log 2>&1; tail -4 gpurun_out/r02/vol_bench2.log
