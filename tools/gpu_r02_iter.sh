#!/bin/bash
# iteration loop of the tile-path kernels: their tests, the whole-model parity tests, an interleaved A/B
O=gpurun_out/r02_iter; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_w4.py tests/test_gpu_golden.py tests/test_gpu_headline.py tests/test_gpu_kernels.py -q -x -p no:cacheprovider 2>&1 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 300 python bench.py --no-extra --cpu-seconds 0 --steps 20 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1])
print('default', d['value'], d['ms_per_step'], d['roofline']['per_layer_avg_ms'], d['parity'])
PY
python tools/ab_kernels.py 8
