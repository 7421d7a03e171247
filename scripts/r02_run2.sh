cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "config2 or device_resident or golden_host" > gpurun_out/r02/t2.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r02/t2.log
python scripts/single_levels.py "" > gpurun_out/r02/single_levels2.log 2>&1; cat gpurun_out/r02/single_levels2.log
rm -rf gpurun_out/r02/trace_single
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02/trace_single -- python3 scripts/single_trace.py > gpurun_out/r02/trace_single.log 2>&1; echo "trace rc=$?"
find gpurun_out/r02/trace_single -name "*kernel_trace.csv" | head
