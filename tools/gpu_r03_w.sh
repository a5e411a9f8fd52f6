# round 3, GPU call W: dense_mid_kernel with exact waits: parity + sweep
O=$GRAFT_REPO_ROOT/gpurun_out/r03w; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mid.py tests/test_gpu_golden.py tests/test_gpu_kernels.py -q -m gpu --timeout 600 > $O/pytest.txt 2>&1; echo "pytest rc $?"
tail -4 $O/pytest.txt | cut -c1-200
timeout 600 python tools/mid_sweep.py 1024 2048 3072 4096 6144 8192 > $O/sweep.txt 2>&1; cat $O/sweep.txt
