cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02/t14.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r02/t14.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r02.json 2> gpurun_out/r02/bench_r02.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_r02.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['single_image'], d['cpu_baseline']['value'])"
bash scripts/profile_gpu.sh r02 > gpurun_out/r02/profile_r02.log 2>&1; tail -8 gpurun_out/r02/profile_r02.log
rm -rf gpurun_out/prof_entries_r02
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_entries_r02 -- python3 scripts/measure_entries.py > gpurun_out/r02/entries_r02.log 2>&1; echo "entries rc=$?"; tail -30 gpurun_out/r02/entries_r02.log
find gpurun_out/prof_r02 gpurun_out/prof_entries_r02 -name "*.csv" -size +4M -delete
