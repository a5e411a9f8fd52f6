for c in 0 8192 16384 32768; do
  for k in 4 2; do
  timeout 200 python bench.py --no-extra --cpu-seconds 0 --steps 15 --warmup 3 --chunk-rows $c --tile-kernel $k > /tmp/b.json 2>/tmp/b.err
  python -c "
import json
d=json.loads(open('/tmp/b.json').read().strip().split('\n')[-1]); print('chunk $c kernel $k', d['value'], d['ms_per_step'], d['roofline']['per_layer_avg_ms'])"
  done
done
