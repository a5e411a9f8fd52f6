#!/bin/bash
# A/B of two builds of the library on one box: tools/gpu_ab.sh  (B = monoloco_amd/lib/libmonoloco_hip_B.so)
cd $GRAFT_REPO_ROOT
L=monoloco_amd/lib
cp $L/libmonoloco_hip.so $L/A.keep
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then cp $L/A.keep $L/libmonoloco_hip.so; else cp $L/libmonoloco_hip_$v.so $L/libmonoloco_hip.so; fi
    python bench.py --steps 30 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['per_layer_avg_ms'])"
  done
done
cp $L/libmonoloco_hip_${TESTV:-B}.so $L/libmonoloco_hip.so
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -2
cp $L/A.keep $L/libmonoloco_hip.so
