# round 3, GPU call AC: single-image frame without copy operations (zero-copy pinned I/O, geometry in the heads launch)
O=$GRAFT_REPO_ROOT/gpurun_out/r03ac; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_kernels.py tests/test_compat.py tests/test_gpu_headline.py -q -m gpu --timeout 600 > $O/pytest.txt 2>&1; echo "pytest rc $?"
tail -4 $O/pytest.txt | cut -c1-200
timeout 300 python tools/latency_loco.py 2>&1 | grep -v amdgpu.ids | tail -6
for rep in 1 2; do
  MONOLOCO_HIP_LIB=$GRAFT_REPO_ROOT/monoloco_amd/lib/libmonoloco_hip_prev.so timeout 300 python tools/latency_loco.py 2>&1 | grep "Loco.forward" | sed 's/^/previous build: /'
  timeout 300 python tools/latency_loco.py 2>&1 | grep "Loco.forward" | sed 's/^/this build:     /'
done
