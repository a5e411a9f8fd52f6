# round 3, GPU call M: dense_mid_kernel parity (f16 mode fix), per-tensor training deviations per route, low-row sweep, kernel stats of the mid path
O=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mid.py -q -m gpu --timeout 600 > $O/pytest_mid.txt 2>&1; echo "pytest mid rc $?"
tail -5 $O/pytest_mid.txt
timeout 600 python tools/exp_train_h1024.py > $O/train_h1024.txt 2>&1; echo "diag rc $?"; cat $O/train_h1024.txt | cut -c1-200
timeout 600 python tools/mid_sweep.py 256 512 768 1024 1536 2048 2560 > $O/sweep_low.txt 2>&1; cat $O/sweep_low.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pm -o stats -- python $GRAFT_REPO_ROOT/tools/mid_sweep.py 4096 8192 > $O/prof_mid.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/pm -name "*.db" | head -1) > $O/mid_kernel_stats.txt 2>&1
head -40 $O/mid_kernel_stats.txt | cut -c1-170
