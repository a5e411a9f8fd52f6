# round 3, GPU call R: xgemm -- requests still outstanding at the end of a wave?
O=$GRAFT_REPO_ROOT/gpurun_out/r03r; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in 64 1 0; do
  L=$GRAFT_REPO_ROOT/monoloco_amd/lib/libmonoloco_hip_xg$v.so; [ $v = 0 ] && L=$GRAFT_REPO_ROOT/monoloco_amd/lib/libmonoloco_hip.so
  REPS=30 MONOLOCO_HIP_LIB=$L timeout 300 python tools/exp_xgemm_occ.py > $O/occ$v.txt 2>&1
  echo "== variant $v: wrong tiles per run: $(grep '^run' $O/occ$v.txt | sed 's/run [0-9]*: \([0-9]*\) of.*/\1/' | tr '\n' ' ')"
done
