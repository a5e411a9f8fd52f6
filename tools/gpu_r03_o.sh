# round 3, GPU call O: xgemm at larger M, where the mid training route drifts
O=$GRAFT_REPO_ROOT/gpurun_out/r03o; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train_mid.py -q -m gpu --timeout 600 -k "xgemm" > $O/pytest_xgemm.txt 2>&1; echo "pytest xgemm rc $?"
grep -n "passed\|failed\|^FAILED\|^E  " $O/pytest_xgemm.txt | head -20
timeout 600 python tools/r03_mid_bringup.py rows 1280 1536 2048 > $O/rows.txt 2>&1; echo "rows rc $?"
grep -n "hidden\| za0 \| a1 " $O/rows.txt | cut -c1-400
