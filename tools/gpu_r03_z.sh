# round 3, GPU call Z: transposed activation lines written by the forward pass (fast training route)
O=$GRAFT_REPO_ROOT/gpurun_out/r03z; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_mid.py tests/test_gpu_headline.py -q -m gpu --timeout 600 > $O/pytest.txt 2>&1; echo "pytest rc $?"
tail -4 $O/pytest.txt | cut -c1-200
timeout 300 python tools/bench_train.py --steps 10 --cpu-seconds 0.1 2>/dev/null | cut -c1-60,200-330
