O=$GRAFT_REPO_ROOT/gpurun_out/prof_train_pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/ptp/sq -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_train.py 3 > $O/log_sq.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/ptp/fetch -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_train.py 3 > $O/log_f.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/ptp/write -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_train.py 3 > $O/log_w.txt 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/ptp -name "*.db") > $O/pmc.txt 2>&1
grep -E "dense_kernel_w4|sgemm|tlines|bn_relu_drop_lines|bn_bwd_fused|bwd_stats" $O/pmc.txt | cut -c1-120 | head -60
