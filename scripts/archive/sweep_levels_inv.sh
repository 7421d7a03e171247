for nb in 4 1; do for sz in 4096 2048 1024 512; do
  echo "== inverse size $sz images $nb"
  INVERSE=1 SIZE=$sz LEVELS=1 IMAGES=$nb ROUNDS=5 STEPS=10 python scripts/sweep.py "" "cpt=4,tile_pairs=64" "cpt=4,tile_pairs=16" "cpt=4,tile_pairs=8" "cpt=4,tile_pairs=4" "cpt=8,tile_pairs=32" "cpt=8,tile_pairs=16" "cpt=8,tile_pairs=8" "cpt=8,tile_pairs=4" "cpt=4,tile_pairs=16,nt_inv=0" 2>&1 | grep -v amdgpu.ids
done; done
