# round 3, GPU call U: whole GPU suite + bench + mid-size sweep + training timing per route
O=$GRAFT_REPO_ROOT/gpurun_out/r03u; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?"
tail -8 $O/pytest_gpu.txt | cut -c1-300
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -2 $O/bench.err
timeout 300 python tools/mid_sweep.py 512 1024 2048 4096 6144 8192 9216 > $O/sweep.txt 2>&1; cat $O/sweep.txt
timeout 300 python tools/r03_mid_bringup.py timing > $O/timing.txt 2>&1; cat $O/timing.txt
