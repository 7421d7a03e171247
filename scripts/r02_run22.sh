cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "ring_layout or tile_variants" > gpurun_out/r02/t22.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02/t22.log
OUT=gpurun_out/r02/sweep_packed.log; : > $OUT
for nb in 1 8; do echo "== images $nb" >> $OUT; IMAGES=$nb ROUNDS=5 STEPS=10 python scripts/sweep.py "nt=7" "nt=15" "nt=7" "nt=15" 2>&1 | grep -v amdgpu >> $OUT; done
python scripts/single_levels.py "nt=7" "nt=15" "nt=7" "nt=15" >> $OUT 2>&1
cat $OUT
