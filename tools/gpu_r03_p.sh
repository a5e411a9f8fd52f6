# round 3, GPU call P: xgemm at > 2 workgroups per CU -- which change makes the 2048 x 1024 x 1024 product right
O=$GRAFT_REPO_ROOT/gpurun_out/r03p; mkdir -p $O; cd $GRAFT_REPO_ROOT
for v in 0 1 2 4; do
  L=$GRAFT_REPO_ROOT/monoloco_amd/lib/libmonoloco_hip_xg$v.so; [ $v = 0 ] && L=$GRAFT_REPO_ROOT/monoloco_amd/lib/libmonoloco_hip.so
  MONOLOCO_HIP_LIB=$L timeout 300 python -m pytest tests/test_gpu_train_mid.py -q -m gpu --timeout 300 -k "xgemm_layouts" > $O/xg$v.txt 2>&1
  echo "variant $v: $(tail -1 $O/xg$v.txt)  $(grep -c '^FAILED' $O/xg$v.txt) failed: $(grep '^FAILED' $O/xg$v.txt | cut -c1-120 | tr '\n' ' ')"
done
