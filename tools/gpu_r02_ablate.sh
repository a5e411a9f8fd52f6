#!/bin/bash
# timing ablations of the w4 kernel: compile-time variants (-DML_W4_ABL=bits) built into monoloco_amd/lib/abl/
O=gpurun_out/r02_ablate2; mkdir -p $O
run() { # tag, lib
  tag=$1; lib=$2
  MONOLOCO_HIP_LIB=$lib timeout 200 python bench.py --no-extra --cpu-seconds 0 --steps 10 --warmup 3 $3 > $O/b_$tag.json 2> $O/b_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/b_$tag.json').read().strip().split('\n')[-1])
    print('%-28s %6.3f ms/step  layers %s' % ('$tag', d['ms_per_step'], d['roofline']['per_layer_avg_ms']))
except Exception as e: print('$tag', 'FAILED', e)
PY
}
L=$GRAFT_REPO_ROOT/monoloco_amd/lib
run w4_full $L/libmonoloco_hip.so
run w4_noepi $L/abl/libml_abl1.so
run w4_nodma $L/abl/libml_abl4.so
run w4_noepi_nodma $L/abl/libml_abl5.so
run w4_noread $L/abl/libml_abl8.so
run w4_nobar $L/abl/libml_abl32.so
run w4_mfma_only $L/abl/libml_abl45.so
run w4_epi_nostore $L/abl/libml_abl64.so
run w4_epi_l2store $L/abl/libml_abl128.so
run pp_full $L/libmonoloco_hip.so "--tile-kernel 2"
