# round 3, last pass over the final tree: smoke(), the whole -m gpu suite, bench.py
O=$GRAFT_REPO_ROOT/gpurun_out/r03final2; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?"
tail -4 $O/pytest_gpu.txt | cut -c1-250
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
