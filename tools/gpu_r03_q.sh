# round 3, GPU call Q: xgemm at > 2 workgroups per CU -- failing tiles, what they miss, more variants
O=$GRAFT_REPO_ROOT/gpurun_out/r03q; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 300 python tools/exp_xgemm_occ.py > $O/occ0.txt 2>&1; cut -c1-900 $O/occ0.txt | grep -v amdgpu.ids
for v in 8 16 32; do
  L=$GRAFT_REPO_ROOT/monoloco_amd/lib/libmonoloco_hip_xg$v.so
  MONOLOCO_HIP_LIB=$L timeout 300 python tools/exp_xgemm_occ.py > $O/occ$v.txt 2>&1
  echo "== variant $v"; grep "^run" $O/occ$v.txt | cut -c1-200
done
