# round 3, GPU call X: xgemm at full occupancy with the accumulators in ArchVGPRs (2) vs AGPRs (4)
O=$GRAFT_REPO_ROOT/gpurun_out/r03x; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in 0; do
  L=$GRAFT_REPO_ROOT/monoloco_amd/lib/libmonoloco_hip.so
  REPS=60 MONOLOCO_HIP_LIB=$L timeout 300 python tools/exp_xgemm_occ.py > $O/occ$v.txt 2>&1
  echo "== variant $v: wrong tiles per run: $(grep '^run' $O/occ$v.txt | sed 's/run [0-9]*: \([0-9]*\) of.*/\1/' | tr '\n' ' ')"
done
MONOLOCO_HIP_LIB=$GRAFT_REPO_ROOT/monoloco_amd/lib/libmonoloco_hip.so timeout 300 python tools/r03_mid_bringup.py timing > $O/timing2.txt 2>&1; cat $O/timing2.txt
timeout 600 python -m pytest tests/test_gpu_train_mid.py tests/test_gpu_train.py -q -m gpu --timeout 600 2>&1 | tail -3
