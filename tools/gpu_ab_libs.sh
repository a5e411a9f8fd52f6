# same-box A/B of two builds of the library through bench.py (value = persons/s), alternating
for i in 1 2 3; do
for l in libmonoloco_hip_prev.so libmonoloco_hip.so; do
echo -n "$l  "; MONOLOCO_HIP_LIB=$PWD/monoloco_amd/lib/$l timeout 200 python bench.py --steps 30 --warmup 5 --no-extra --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('per_layer_avg_ms'))"
done; done
