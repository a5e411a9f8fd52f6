#!/bin/bash
# rocprofv3 kernel stats + PMC passes of bench.py (run through gpurun from the repo root): tools/gpu_profile.sh <tag>
TAG=${1:-run}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd $R; timeout 300 python bench.py --steps 30 --warmup 5 > $O/bench.json 2>$O/bench.err
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- $B > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq -o pmc -- $B --no-profile > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/pmc_lds -o pmc -- $B --no-profile > $O/pmc_lds.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc -- $B --no-profile > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc -- $B --no-profile > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_tcc -o pmc -- $B --no-profile > $O/pmc_tcc.log 2>&1
tail -1 $O/bench.json | cut -c1-400
