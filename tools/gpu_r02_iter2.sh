#!/bin/bash
O=gpurun_out/r02_iter; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_w4.py -q -x -p no:cacheprovider 2>&1 ) > $O/pytest.log 2>&1
tail -2 $O/pytest.log
python tools/ab_kernels.py 8
export MONOLOCO_HIP_LIB=$GRAFT_REPO_ROOT/monoloco_amd/lib/libmonoloco_hip_trace.so
rm -f /tmp/trace.bin
ML_DENSE_TRACE=/tmp/trace.bin ML_TILE_KERNEL=4 timeout 200 python bench.py --no-extra --cpu-seconds 0 --steps 1 --warmup 1 --no-profile > /tmp/b.json 2>/tmp/b.err
python tools/trace_summary_w4.py /tmp/trace.bin 16 | tail -14 | cut -c1-200
