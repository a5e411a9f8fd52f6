#!/bin/bash
# Round 5: same-box A/B of two builds of the library (A = monoloco_amd/lib/libmonoloco_hip_prev.so, B = libmonoloco_hip.so):
#   tools/gpu_ab_r5.sh <tag> [pmc]     -> gpurun_out/<tag>/ab.txt (+ pmc_A.txt / pmc_B.txt with the LDS and TCC passes when `pmc` is given)
# alternating short bench runs (ms per step + per-layer ms from HIP events), then per build one rocprofv3 pass each of the LDS
# counters, FETCH_SIZE and WRITE_SIZE of `bench.py --steps 6 --warmup 2`.
TAG=${1:?tag}; PMC=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
L=$R/monoloco_amd/lib
B="python $R/bench.py --no-extra --cpu-seconds 0 --no-live-counters --no-parity"
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then lib=$L/libmonoloco_hip_prev.so; else lib=$L/libmonoloco_hip.so; fi
    echo -n "$v  " >> $O/ab.txt
    MONOLOCO_HIP_LIB=$lib timeout 200 $B --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('per_layer_avg_ms'))" >> $O/ab.txt
  done
done
cat $O/ab.txt
if [ -n "$PMC" ]; then
  cd /tmp; export TMPDIR=/tmp
  for v in A B; do
    if [ $v = A ]; then lib=$L/libmonoloco_hip_prev.so; else lib=$L/libmonoloco_hip.so; fi
    for pass in "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
      set -- $pass; n=$1; shift
      MONOLOCO_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/ab_${v}_$n -o pmc -- $B --steps 6 --warmup 2 --no-profile > $O/pmc_${v}_$n.log 2>&1
    done
    python $R/tools/pmc_summary.py $(ls -d /tmp/ab_${v}_*/ | sed 's#/$##' | xargs -I{} find {} -name "*.db") > $O/pmc_$v.txt 2>&1
    grep "dense_kernel" $O/pmc_$v.txt | grep -E "LDS|FETCH|WRITE" | cut -c1-140
  done
fi
