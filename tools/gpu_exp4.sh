L=$GRAFT_REPO_ROOT/monoloco_amd/lib
for v in libmonoloco_hip.so abl/libml_abl512.so abl/libml_abl1024.so abl/libml_abl2048.so libmonoloco_hip.so; do
  MONOLOCO_HIP_LIB=$L/$v timeout 200 python bench.py --no-extra --cpu-seconds 0 --steps 20 --warmup 3 > /tmp/b.json 2>/tmp/b.err
  python -c "
import json
d=json.loads(open('/tmp/b.json').read().strip().split('\n')[-1]); print('$v', d['ms_per_step'], d['roofline']['per_layer_avg_ms'], d['parity']['max_abs_raw'])"
done
