# round 3, GPU call J: mid route on the exact-fp32 xgemm
O=$GRAFT_REPO_ROOT/gpurun_out/r03j; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train_mid.py tests/test_gpu_train.py -q -m gpu --timeout 300 > $O/pytest_train.txt 2>&1; echo "pytest rc $?"
tail -15 $O/pytest_train.txt
timeout 300 python tools/r03_mid_bringup.py compare > $O/compare.txt 2>&1; echo "compare rc $?"
grep -v "^   " $O/compare.txt | tail -5
timeout 300 python tools/r03_mid_bringup.py timing > $O/timing.txt 2>&1; cat $O/timing.txt
cd /tmp; export TMPDIR=/tmp
for R in 331 512; do
  rm -rf /tmp/pt$R
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt$R -o stats -- python $GRAFT_REPO_ROOT/tools/prof_train.py 20 $R > $O/prof$R.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/pt$R -name "*.db" | head -1) > $O/train_kernel_stats_rows$R.txt 2>&1
done
head -32 $O/train_kernel_stats_rows331.txt | cut -c1-175
tail -4 $O/train_kernel_stats_rows331.txt | cut -c1-175
