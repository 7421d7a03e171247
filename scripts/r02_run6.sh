cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/bench_fuse2.log
: > $OUT
for opt in "fuse2=0" "fuse2=32" "fuse2=64" "fuse2=0" "fuse2=32"; do
  for nb in 64 8; do
    echo "== images $nb $opt" >> $OUT
    python bench.py --steps 20 --warmup 5 --images $nb --no-cpu --no-single --opt $opt 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'], 'step min/med', d['step_ms_rank0']['min'], d['step_ms_rank0']['median'], 'L0(+L1) ms', d['roofline']['avg_launch_ms'])" >> $OUT
  done
done
cat $OUT
