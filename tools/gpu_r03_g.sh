O=$GRAFT_REPO_ROOT/gpurun_out/r03g; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 300 python tools/r03_xgemm_ab.py > $O/xgemm_ablate.txt 2>&1; cat $O/xgemm_ablate.txt
