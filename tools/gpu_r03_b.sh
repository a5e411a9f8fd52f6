# round 3, GPU call B: where does the mid-route GEMM's time go (ablations + PMC)
O=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 300 python tools/r03_tgemm_ablate.py > $O/ablate.txt 2>&1; cat $O/ablate.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE -d /tmp/pmc1 -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_train.py 3 331 > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d /tmp/pmc2 -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_train.py 3 331 > $O/pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d /tmp/pmc3 -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_train.py 3 331 > $O/pmc3.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 -name "*.db") > $O/pmc.txt 2>&1
grep -E "tgemm|apply|adam_tile" $O/pmc.txt | cut -c1-130
tail -3 $O/pmc1.log $O/pmc2.log $O/pmc3.log
