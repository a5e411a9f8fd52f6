# round 3, GPU call AA: xgemm_pair_kernel (data + weight gradient of a Linear in one launch)
O=$GRAFT_REPO_ROOT/gpurun_out/r03aa; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_mid.py -q -m gpu --timeout 600 > $O/pytest.txt 2>&1; echo "pytest rc $?"
tail -4 $O/pytest.txt | cut -c1-200
timeout 300 python tools/r03_mid_bringup.py timing > $O/timing.txt 2>&1; cat $O/timing.txt
