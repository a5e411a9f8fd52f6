# round 3, GPU call K: whole GPU suite + bench line + the rows-331/512 training profiles of the same tree
O=$GRAFT_REPO_ROOT/gpurun_out/r03k; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?"
tail -8 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -3 $O/bench.err
cd /tmp; export TMPDIR=/tmp
for R in 331 512; do
  rm -rf /tmp/pt$R
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt$R -o stats -- python $GRAFT_REPO_ROOT/tools/prof_train.py 20 $R > $O/prof$R.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/pt$R -name "*.db" | head -1) > $O/train_kernel_stats_rows$R.txt 2>&1
done
tail -4 $O/train_kernel_stats_rows331.txt | cut -c1-175
