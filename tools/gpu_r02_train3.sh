O=gpurun_out/r02_train3; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_train.py -q -p no:cacheprovider 2>&1 ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python tools/bench_train.py --steps 20 --cpu-seconds 0.1 2>&1 | cut -c1-60,230-330 | tee $O/bench_train.log
