#!/bin/bash
# round 2, first GPU pass: full -m gpu suite, the bench line, kernel-trace stats
O=gpurun_out/r02_run2; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider 2>&1 ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
( time timeout 400 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.err
python - <<'PY'
import json
l=open('gpurun_out/r02_run2/bench.json').read().strip().split('\n')[-1]
try:
    d=json.loads(l); print({k:d[k] for k in ('value','ms_per_step','parity','e2e_ms') if k in d}); print(d.get('roofline',{}).get('per_layer_avg_ms')); print(json.dumps(d.get('extra'))[:3000]); print(d.get('cpu_baseline'))
except Exception as e: print('bad bench line', e, l[:500])
PY
