#!/bin/bash
# iteration loop of the w4 kernel: its tests, an A/B bench, a timeline
O=gpurun_out/r02_iter; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_w4.py -q -x -p no:cacheprovider 2>&1 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for K in 4 260 2; do
  timeout 300 python bench.py --tile-kernel $K --no-extra --cpu-seconds 0 --steps 20 > $O/bench_k$K.json 2> $O/bench_k$K.err
  python - <<PY
import json
d=json.loads(open('$O/bench_k$K.json').read().strip().split('\n')[-1])
print('kernel $K', d['value'], d['ms_per_step'], d['roofline']['per_layer_avg_ms'], d['parity']['max_abs_xyzds'])
PY
done
export MONOLOCO_HIP_LIB=$GRAFT_REPO_ROOT/monoloco_amd/lib/libmonoloco_hip_trace.so
rm -f /tmp/trace.bin
ML_DENSE_TRACE=/tmp/trace.bin ML_TILE_KERNEL=4 timeout 200 python bench.py --no-extra --cpu-seconds 0 --steps 1 --warmup 1 --no-profile > /tmp/b.json 2>/tmp/b.err
python tools/trace_summary_w4.py /tmp/trace.bin 16 | tail -22 | cut -c1-200
