# sample board power / clocks while bench.py runs (evidence for the "clock is set by power" statement in profiles/r02_ablation.md)
O=gpurun_out/r02_power; mkdir -p $O
rocm-smi --showmaxpower --showpower --showclocks --showtemp > $O/idle.txt 2>&1
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showtemp --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 0.25; done ) > $O/samples.txt 2>&1 &
SP=$!
timeout 200 python bench.py --steps 8000 --warmup 20 --no-extra --cpu-seconds 0 --no-profile 2>&1 | tail -1 | cut -c1-260 > $O/bench_long.txt
kill $SP 2>/dev/null; wait $SP 2>/dev/null
head -30 $O/idle.txt; sed -n 1,3p $O/samples.txt; sed -n 20,40p $O/samples.txt | cut -c1-300; cat $O/bench_long.txt
