cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "small_levels or tile_variants or device_resident or config" > gpurun_out/r02/t15.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r02/t15.log
python scripts/single_levels.py "" "tail=0" > gpurun_out/r02/single_levels4.log 2>&1; cat gpurun_out/r02/single_levels4.log
