O=gpurun_out/r02_run3; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python tools/latency.py 1 16 64 128 256 > $O/latency.txt 2>&1; grep rows $O/latency.txt
python tools/latency_loco.py > $O/lat.txt 2>&1; tail -1 $O/lat.txt
bash tools/gpu_ab_libs.sh
