cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/sweep_fuse2_64.log
: > $OUT
echo "== 8192^2 J=5, 64 images" >> $OUT
IMAGES=64 ROUNDS=3 STEPS=3 timeout -k 10 500 python scripts/sweep.py "fuse2=0" "fuse2=64" "fuse2=32" "fuse2=128" "fuse2=64,fuse2_pc=0" "fuse2=64,ring=8" 2>&1 | grep -v amdgpu.ids >> $OUT
echo "== 8192^2 J=2 (levels 0+1 only), 64 images" >> $OUT
IMAGES=64 LEVELS=2 ROUNDS=3 STEPS=3 timeout -k 10 500 python scripts/sweep.py "fuse2=0" "fuse2=64" 2>&1 | grep -v amdgpu.ids >> $OUT
echo "== 4096^2 J=5, 32 images (config 4 per GPU)" >> $OUT
SIZE=4096 IMAGES=32 ROUNDS=3 STEPS=5 timeout -k 10 500 python scripts/sweep.py "fuse2=0" "fuse2=64" "fuse2=32" 2>&1 | grep -v amdgpu.ids >> $OUT
echo "== 4096^2 J=5, 256 images (config 4 on one GPU)" >> $OUT
SIZE=4096 IMAGES=256 ROUNDS=3 STEPS=3 timeout -k 10 500 python scripts/sweep.py "fuse2=0" "fuse2=64" "fuse2=32" 2>&1 | grep -v amdgpu.ids >> $OUT
cat $OUT
