O=gpurun_out/r02_lat; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_compat.py tests/test_gpu_train.py -q -p no:cacheprovider 2>&1 ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python tools/latency_loco.py > $O/lat.txt 2>&1; tail -1 $O/lat.txt
