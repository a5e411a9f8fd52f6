# round 3, GPU call V: kernel stats + SQ / LDS counters of the mid-size path (bench.py --batch 4096 / 8192)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for BATCH in 4096 8192; do
  O=$R/gpurun_out/prof_r03mid$BATCH; mkdir -p $O; W=/tmp/prof_mid$BATCH; mkdir -p $W
  B="python $R/bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-extra --batch $BATCH"
  timeout 300 rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- $B > $O/stats.log 2>&1
  python $R/tools/rocprof_summary.py $(find $W/stats -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
  pmc() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $W/pmc_$n -o pmc -- $B --no-profile > $O/pmc_$n.log 2>&1; }
  pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
  pmc sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
  pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
  python $R/tools/pmc_summary.py $(ls -d $W/pmc_*/ | sed 's#/$##' | xargs -I{} find {} -name "*.db") > $O/pmc.txt 2>&1
  echo "== batch $BATCH"; head -14 $O/kernel_stats.txt | cut -c1-150
  grep "dense_mid" $O/pmc.txt | cut -c1-130
done
