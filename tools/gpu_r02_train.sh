O=gpurun_out/r02_train; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_headline.py -q -x -k "train or config5" -p no:cacheprovider 2>&1 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python tools/bench_train.py --steps 6 --cpu-seconds 0.1 2>&1 | cut -c1-260
