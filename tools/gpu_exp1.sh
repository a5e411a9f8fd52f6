#!/bin/bash
# ablation + PMC collection for the dense kernel (run through gpurun from the repo root)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp1; mkdir -p $O
cd $R
for dbg in 0 1 2 4 5; do
  ML_DENSE_DEBUG=$dbg timeout 200 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 > $O/bench_dbg$dbg.json 2>$O/bench_dbg$dbg.err
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq -o pmc_sq -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-profile > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/pmc_lds -o pmc_lds -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-profile > $O/pmc_lds.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-profile > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc_write -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-profile > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $O/pmc_tcc -o pmc_tcc -- python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-profile > $O/pmc_tcc.log 2>&1
ls $O $O/*/ | head -40
for f in $O/bench_dbg*.json; do echo $f; python -c "
import json,sys
l=open('$f').read().strip().splitlines()
try:
  d=json.loads(l[-1]); print(d['value'], d['ms_per_step'], d['roofline']['per_layer_avg_ms'])
except Exception as e: print('ERR', e, l[-3:])
"; done
