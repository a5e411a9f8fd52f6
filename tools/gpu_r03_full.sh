# round 3: the whole GPU suite + the bench line (what the driver runs at round end)
O=$GRAFT_REPO_ROOT/gpurun_out/r03full; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu --timeout 600 > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?"
tail -25 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -3 $O/bench.err
python - <<'PY'
import json,os
p=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/r03full/bench.json')
try:
    d=json.loads(open(p).read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d.get('roofline',{}).get('frac'))
    ex=d.get('extra',{})
    for k in ('e2e','train','train_epoch','latency_16_persons','config4_1M_rows_one_gpu','stereo_32768'):
        print(k, json.dumps(ex.get(k))[:700])
except Exception as e:
    print('bench parse failed', e)
PY
