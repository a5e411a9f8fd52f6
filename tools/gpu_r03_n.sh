# round 3, GPU call N: mid training route vs exact route as the batch grows; dense_mid_kernel v2 (barrier inside the step) parity + sweeps
O=$GRAFT_REPO_ROOT/gpurun_out/r03n; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 600 python tools/r03_mid_bringup.py rows 1024 2048 3072 4096 > $O/rows.txt 2>&1; echo "rows rc $?"
grep -v "grad .*bias\|batch_norm" $O/rows.txt | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_mid.py -q -m gpu --timeout 600 -x > $O/pytest_mid.txt 2>&1; echo "pytest mid rc $?"
tail -3 $O/pytest_mid.txt
timeout 600 python tools/mid_sweep.py 512 1024 1536 2048 3072 4096 6144 8192 12288 16384 > $O/sweep.txt 2>&1; cat $O/sweep.txt
