# same-box A/B: does the timed step depend on the VALUES (power is data-dependent)?  reference-trained weights + real poses vs seeded synthetic
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do for w in reference synthetic; do
echo -n "$w  "; timeout 200 python bench.py --steps 30 --warmup 5 --no-extra --cpu-seconds 0 --weights $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['power']['socket_power_w_median'], d['roofline']['power'].get('sclk_mhz_median'))"
done; done
