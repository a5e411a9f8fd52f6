# round 3, GPU call AD: dense_mid_kernel with 256 x 128 tiles (one workgroup per CU)
O=$GRAFT_REPO_ROOT/gpurun_out/r03ad; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mid.py -q -m gpu --timeout 600 -x > $O/pytest.txt 2>&1; echo "pytest rc $?"
tail -4 $O/pytest.txt | cut -c1-200
timeout 600 python tools/mid_sweep.py 4096 6144 8192 12288 16384 32768 > $O/sweep.txt 2>&1; cat $O/sweep.txt
