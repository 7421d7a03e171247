for sz in 8192 4096 2048 1024 512; do
  echo "== size $sz (1 level, 4 images)"
  SIZE=$sz LEVELS=1 IMAGES=4 ROUNDS=5 STEPS=10 python scripts/sweep.py "cpt=8,tile_pairs=64" "cpt=8,tile_pairs=32" "cpt=8,tile_pairs=16" "cpt=8,tile_pairs=8" "cpt=8,tile_pairs=4" "cpt=4,tile_pairs=64" "cpt=4,tile_pairs=32" "cpt=4,tile_pairs=16" "cpt=4,tile_pairs=8" "cpt=4,tile_pairs=4" "cpt=4,tile_pairs=8,waves=2" "cpt=4,tile_pairs=8,waves=1" 2>&1 | grep -v amdgpu.ids
done
