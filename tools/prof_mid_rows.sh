# kernel stats of `bench.py --batch <rows>` inside the mid window (through gpurun): bash tools/prof_mid_rows.sh <out tag> [rows ...]
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=${1:-mid}; shift; mkdir -p $R/gpurun_out/$TAG
for B in ${@:-4096 2048 1024}; do
rm -rf /tmp/pm$B
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pm$B -o stats -- python $R/bench.py --batch $B --steps 200 --warmup 20 --cpu-seconds 0 --no-extra --no-profile --no-parity --no-live-counters > $R/gpurun_out/$TAG/bench$B.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/pm$B -name "*.db" | head -1) > $R/gpurun_out/$TAG/kernel_stats_rows$B.txt 2>&1
head -9 $R/gpurun_out/$TAG/kernel_stats_rows$B.txt | cut -c1-160
done
