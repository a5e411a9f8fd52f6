# Round 6: does removing idle matrix-pipe cycles from the headline kernel return as time, or as a lower clock?  The full library against two
# compile-time ablations of dense_kernel_w4 (-DML_W4_ABL=64: the epilogue without its global stores; =1: no epilogue at all -- results are
# garbage, only time / counters / power are read), each through `bench.py` with its live rocprofv3 counters (SQ_VALU_MFMA_BUSY_CYCLES,
# GRBM_GUI_ACTIVE) and its power probe (rocm_smi: socket power, sclk), alternating, two rounds.   bash tools/w4_power_probe.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-w4power}; mkdir -p $O; cd $R
for ROUND in 1 2; do
  for V in full w4abl64 w4abl1; do
    LIB=$R/monoloco_amd/lib/libmonoloco_hip.so; [ $V != full ] && LIB=$R/monoloco_amd/lib/libmonoloco_hip_$V.so
    MONOLOCO_HIP_LIB=$LIB timeout 600 python bench.py --steps 200 --warmup 20 --no-extra --cpu-seconds 0 --no-parity > $O/${V}_$ROUND.json 2> $O/${V}_$ROUND.err
    python - "$O/${V}_$ROUND.json" $V <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']; p = r.get('power') or {}
    print("%-8s ms/step %.4f  value %.2f M  mfma_busy %s  dominant kernel us %s  power %s W  sclk %s MHz  per-layer ms %s" % (
        sys.argv[2], d['ms_per_step'], d['value'] / 1e6, r.get('mfma_busy'), r.get('dominant_kernel_avg_us_live'),
        p.get('socket_power_w_median'), p.get('sclk_mhz_median'), r.get('per_layer_avg_ms')))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
  done
done
