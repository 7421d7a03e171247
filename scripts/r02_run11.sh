cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_hip_volume.py -x -q -m gpu > gpurun_out/r02/t11.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02/t11.log
VARIANTS="vol_fused=1;vol_fused=1,vol_nt=7;vol_fused=1,vol_nt=2;vol_fused=1;vol_fused=1,vol_nt=7" python scripts/vol_op_bench.py 1024 3 > gpurun_out/r02/vol_bench.log 2>&1; cat gpurun_out/r02/vol_bench.log
python scripts/single_levels.py "" > gpurun_out/r02/single_levels3.log 2>&1; cat gpurun_out/r02/single_levels3.log
