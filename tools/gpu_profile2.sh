#!/bin/bash
# rocprofv3 kernel stats + PMC passes of bench.py, summaries only (run through gpurun from the repo root):
#   tools/gpu_profile2.sh <tag> [extra bench.py arguments]
TAG=${1:-run}; shift; EXTRA="$@"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O; W=/tmp/prof_$TAG; mkdir -p $W
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --cpu-seconds 0 --no-extra --no-live-counters --no-parity $EXTRA"
timeout 300 rocprofv3 --kernel-trace --stats -d $W/stats -o stats -- $B > $O/stats.log 2>&1
python $R/tools/rocprof_summary.py $(find $W/stats -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
pmc() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $W/pmc_$n -o pmc -- $B --no-profile > $O/pmc_$n.log 2>&1; }
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pmc sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum
python $R/tools/pmc_summary.py $(ls -d $W/pmc_*/ | sed 's#/$##' | xargs -I{} find {} -name "*.db") > $O/pmc.txt 2>&1
grep -h "dense_kernel" $O/kernel_stats.txt | head -8
grep "dense_kernel" $O/pmc.txt | head -80
