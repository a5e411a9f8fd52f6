# round 3, final GPU pass: smoke(), the whole -m gpu suite, bench.py, kernel stats + PMC passes of the bench command, training profiles
O=$GRAFT_REPO_ROOT/gpurun_out/r03final; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?"
tail -5 $O/pytest_gpu.txt | cut -c1-250
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
bash tools/gpu_profile2.sh r03final > $O/profile2.log 2>&1; echo "profile rc $?"
cp $GRAFT_REPO_ROOT/gpurun_out/prof_r03final/kernel_stats.txt $O/kernel_stats.txt; cp $GRAFT_REPO_ROOT/gpurun_out/prof_r03final/pmc.txt $O/pmc.txt
cd /tmp; export TMPDIR=/tmp
for R in 331 512; do
  rm -rf /tmp/pt$R
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt$R -o stats -- python $GRAFT_REPO_ROOT/tools/prof_train.py 20 $R > $O/prof$R.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/pt$R -name "*.db" | head -1) > $O/train_kernel_stats_rows$R.txt 2>&1
done
head -16 $O/train_kernel_stats_rows331.txt | cut -c1-150; tail -2 $O/train_kernel_stats_rows331.txt | cut -c1-170
head -14 $O/kernel_stats.txt | cut -c1-150
