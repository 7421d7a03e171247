set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python -m pytest tests -x -q -m gpu > gpurun_out/r02/gpu_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r02/gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02/bench_n1.json 2> gpurun_out/r02/bench_n1.err; echo "bench rc=$?"
cat gpurun_out/r02/bench_n1.json
python bench.py --gpus 2 --steps 5 --warmup 2 --images 16 > gpurun_out/r02/bench_n2_shared.json 2> gpurun_out/r02/bench_n2.err; echo "bench2 rc=$?"
cat gpurun_out/r02/bench_n2_shared.json; tail -5 gpurun_out/r02/bench_n2.err
python bench.py --gpus 1 --steps 20 --warmup 5 --images 8 --no-cpu --no-single > gpurun_out/r02/bench_n1_b8.json 2>/dev/null; cat gpurun_out/r02/bench_n1_b8.json
python scripts/single_levels.py "" "tile_pairs=32" "ring=16" "ring=16,tile_pairs=32" "tile_pairs=16" "waves=2" "waves=2,tile_pairs=32" > gpurun_out/r02/single_levels.log 2>&1; cat gpurun_out/r02/single_levels.log
