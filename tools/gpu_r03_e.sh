# round 3, GPU call E: xgemm A/B (rotation, grid order, padded strides) + tests
O=$GRAFT_REPO_ROOT/gpurun_out/r03e; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 300 python tools/r03_xgemm_ab.py > $O/xgemm_ab.txt 2>&1; cat $O/xgemm_ab.txt
timeout 600 python -m pytest tests/test_gpu_train_mid.py tests/test_gpu_train.py -q -m gpu --timeout 300 > $O/pytest_train.txt 2>&1; echo "pytest rc $?"
tail -8 $O/pytest_train.txt
