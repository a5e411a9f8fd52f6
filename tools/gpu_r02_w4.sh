#!/bin/bash
# w4 kernel bring-up: its own tests first, then the whole suite, then an A/B bench
O=gpurun_out/r02_w4; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_w4.py -q -x -p no:cacheprovider 2>&1 ) > $O/pytest_w4.log 2>&1
tail -15 $O/pytest_w4.log
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_w4.py 2>&1 ) > $O/pytest_all.log 2>&1
tail -8 $O/pytest_all.log
for K in 4 2; do
  timeout 300 python bench.py --tile-kernel $K --no-extra --cpu-seconds 0 --steps 20 > $O/bench_k$K.json 2> $O/bench_k$K.err
  python - <<PY
import json
d=json.loads(open('$O/bench_k$K.json').read().strip().split('\n')[-1])
print('kernel $K', d['value'], d['ms_per_step'], d['roofline']['per_layer_avg_ms'], d['parity']['max_abs_xyzds'])
PY
done
