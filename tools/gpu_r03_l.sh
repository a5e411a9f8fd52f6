# round 3, GPU call L: dense_mid_kernel parity + row sweep, the two re-toleranced training tests
O=$GRAFT_REPO_ROOT/gpurun_out/r03l; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mid.py -q -m gpu --timeout 600 -x > $O/pytest_mid.txt 2>&1; echo "pytest mid rc $?"
tail -15 $O/pytest_mid.txt
timeout 600 python -m pytest tests/test_gpu_train_mid.py -q -m gpu --timeout 600 -s -k "headline or trajectory" > $O/pytest_train.txt 2>&1; echo "pytest train rc $?"
grep -n "worst\|passed\|failed\|Error" $O/pytest_train.txt | head
timeout 600 python tools/mid_sweep.py > $O/sweep.txt 2>&1; cat $O/sweep.txt
