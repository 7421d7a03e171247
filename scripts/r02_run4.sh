cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
OUT=gpurun_out/r02/sweep_fuse2.log
: > $OUT
for nb in 8 32; do
echo "== 8192^2 J=5, $nb images" >> $OUT
IMAGES=$nb ROUNDS=4 STEPS=5 timeout -k 10 400 python scripts/sweep.py "fuse2=0" "fuse2=1" "fuse2=64" "fuse2=32" "fuse2=128" "fuse2=64,ring=12" "fuse2=64,ring=10" "fuse2=64,ring=8" "fuse2=32,ring=8" "fuse2=64,fuse2_pc=0" "fuse2=64,nt=0" 2>&1 | grep -v amdgpu.ids >> $OUT
done
echo "== 8192^2 levels 2 only (L0+L1), 8 images" >> $OUT
IMAGES=8 LEVELS=2 ROUNDS=4 STEPS=5 timeout -k 10 300 python scripts/sweep.py "fuse2=0" "fuse2=64" "fuse2=32" "fuse2=64,fuse2_pc=0" 2>&1 | grep -v amdgpu.ids >> $OUT
cat $OUT
