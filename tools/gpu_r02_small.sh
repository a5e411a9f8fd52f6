O=gpurun_out/r02_small; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_headline.py -q -p no:cacheprovider 2>&1 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
python tools/latency.py 1 16 64 128 256 1024 2048 > $O/latency.txt 2>&1; grep rows $O/latency.txt
