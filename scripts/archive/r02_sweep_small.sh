# single-image, single-level sweeps of tile shape per level size (what the tails of one 8192^2 call see)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/sweep_small.log
: > $OUT
for sz in 4096 2048 1024 512 256; do
  echo "== size $sz (1 level, 1 image)" >> $OUT
  V=""
  for cpt in 8 4; do for tp in 64 32 16 8 4 2; do for ring in 8 16; do V="$V cpt=$cpt,tile_pairs=$tp,ring=$ring"; done; done; done
  V="$V cpt=4,tile_pairs=4,ring=8,waves=2 cpt=4,tile_pairs=4,ring=8,waves=1 cpt=4,tile_pairs=2,ring=8,waves=2 cpt=4,tile_pairs=8,ring=16,waves=2 cpt=4,tile_pairs=8,ring=16,waves=1"
  SIZE=$sz LEVELS=1 IMAGES=1 ROUNDS=4 STEPS=10 python scripts/sweep.py $V 2>&1 | grep -v amdgpu.ids >> $OUT
done
for sz in 4096 2048 1024 512; do
  echo "== size $sz (1 level, 8 images)" >> $OUT
  V=""
  for cpt in 8 4; do for tp in 64 32 16 8 4; do for ring in 8 16; do V="$V cpt=$cpt,tile_pairs=$tp,ring=$ring"; done; done; done
  SIZE=$sz LEVELS=1 IMAGES=8 ROUNDS=4 STEPS=10 python scripts/sweep.py $V 2>&1 | grep -v amdgpu.ids >> $OUT
done
tail -5 $OUT
