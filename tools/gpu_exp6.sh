export MONOLOCO_HIP_LIB=$GRAFT_REPO_ROOT/monoloco_amd/lib/libmonoloco_hip_trace.so
rm -f /tmp/trace.bin
ML_DENSE_TRACE=/tmp/trace.bin ML_TILE_KERNEL=4 timeout 200 python bench.py --no-extra --cpu-seconds 0 --steps 2 --warmup 1 --no-profile > /tmp/b.json 2>/tmp/b.err
tail -3 /tmp/b.err
python tools/trace_summary_w4.py /tmp/trace.bin 16 | tail -42
