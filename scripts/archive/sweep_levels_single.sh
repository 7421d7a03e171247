for sz in 8192 4096 2048; do
  echo "== forward size $sz images 1"
  SIZE=$sz LEVELS=1 IMAGES=1 ROUNDS=5 STEPS=10 python scripts/sweep.py "" "cpt=8,tile_pairs=64" "cpt=8,tile_pairs=32" "cpt=8,tile_pairs=16" "cpt=8,tile_pairs=8" "cpt=8,tile_pairs=4" "cpt=4,tile_pairs=32" "cpt=4,tile_pairs=16" "cpt=4,tile_pairs=8" "cpt=8,tile_pairs=32,ring=16,wave_horiz=1" "cpt=8,tile_pairs=16,ring=16,wave_horiz=1" 2>&1 | grep -v amdgpu.ids
done
