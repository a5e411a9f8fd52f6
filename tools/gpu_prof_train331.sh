O=$GRAFT_REPO_ROOT/gpurun_out/prof_train331; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt3 -o stats -- python $GRAFT_REPO_ROOT/tools/prof_train.py 20 331 > $O/log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/pt3 -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
head -40 $O/kernel_stats.txt | cut -c1-170
