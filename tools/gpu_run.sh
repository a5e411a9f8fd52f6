#!/bin/bash
# One parameterised GPU pass (run through gpurun from the repo root); replaces the per-call lease scripts of rounds 1-3.
#   gpurun --timeout T -- 'bash tools/gpu_run.sh <tag> <step> [<step> ...]'
# Everything lands in gpurun_out/<tag>/.  Steps (run in the order given):
#   smoke                       __graft_entry__.smoke()
#   pytest[:<files,comma>]      the -m gpu suite (or only the named files under tests/)
#   bench[:<bench.py args>]     python bench.py <args>   -> bench.json (args with '+' for spaces)
#   prof[:<bench.py args>]      rocprofv3 kernel stats + the separate PMC passes of the bench command (tools/gpu_profile2.sh)
#   trainprof:<rows>[:route]    rocprofv3 kernel stats of training steps at that batch size (tools/prof_train.py)
#   trainpmc:<rows>             FETCH_SIZE / WRITE_SIZE / SQ passes of the same training steps
#   py:<script>[:args]          python tools/<script> <args>  -> <script>.txt  (args with '+' for spaces)
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export PYTHONUNBUFFERED=1
for STEP in "$@"; do
  KIND=${STEP%%:*}; ARG=""; [ "$KIND" != "$STEP" ] && ARG=${STEP#*:}; ARG=${ARG//+/ }
  echo "=== $STEP"
  case $KIND in
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 ;;
    pytest) FILES="tests"; [ -n "$ARG" ] && FILES=$(echo $ARG | tr ',' '\n' | sed 's#^#tests/#' | tr '\n' ' ')
            timeout 1500 python -m pytest $FILES -q -rP -m gpu --timeout 600 > $O/pytest_${ARG//[^a-zA-Z0-9]/_}.txt 2>&1; echo "pytest rc $?"
            tail -6 $O/pytest_${ARG//[^a-zA-Z0-9]/_}.txt | cut -c1-250 ;;
    bench)  N=$(ls $O/bench*.json 2>/dev/null | wc -l); timeout 900 python bench.py $ARG > $O/bench$N.json 2> $O/bench$N.err; echo "bench rc $?"
            python - "$O/bench$N.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value %.4g %s  ms/step %.4f  frac %s" % (d['value'], d['unit'], d['ms_per_step'], d.get('roofline', {}).get('frac')))
    ex = d.get('extra', {})
    for k in ('other_batches', 'train', 'latency_16_persons', 'stereo_32768'):
        if k in ex: print(k, json.dumps(ex[k])[:900])
except Exception as e:
    print("no bench line:", e)
PY
            ;;
    prof)   bash tools/gpu_profile2.sh $TAG $ARG > $O/profile.log 2>&1; echo "prof rc $?"
            cp $R/gpurun_out/prof_$TAG/kernel_stats.txt $R/gpurun_out/prof_$TAG/pmc.txt $O/ 2>/dev/null; head -14 $O/kernel_stats.txt | cut -c1-160 ;;
    trainprof) ROWS=${ARG%%:*}; ROUTE=""; [ "$ROWS" != "$ARG" ] && ROUTE=${ARG#*:}
            ( cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pt$ROWS
              timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pt$ROWS -o stats -- python $R/tools/prof_train.py 12 $ROWS $ROUTE > $O/trainprof$ROWS.log 2>&1
              python $R/tools/rocprof_summary.py $(find /tmp/pt$ROWS -name "*.db" | head -1) > $O/train_kernel_stats_rows$ROWS.txt 2>&1 )
            head -30 $O/train_kernel_stats_rows$ROWS.txt | cut -c1-170; tail -2 $O/train_kernel_stats_rows$ROWS.txt | cut -c1-170 ;;
    trainpmc) ROWS=$ARG
            ( cd /tmp; export TMPDIR=/tmp
              for P in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
                set -- $P; n=$1; shift; rm -rf /tmp/tp_$n
                timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/tp_$n -o pmc -- python $R/tools/prof_train.py 6 $ROWS > $O/trainpmc_$n.log 2>&1
              done
              python $R/tools/pmc_summary.py $(find /tmp/tp_fetch /tmp/tp_write /tmp/tp_sq -name "*.db") > $O/train_pmc_rows$ROWS.txt 2>&1 )
            head -40 $O/train_pmc_rows$ROWS.txt | cut -c1-200 ;;
    py)     S=${ARG%% *}; A=""; [ "$S" != "$ARG" ] && A=${ARG#* }; S2=${S%%:*}; [ "$S2" != "$S" ] && { A="${S#*:} $A"; S=$S2; }
            timeout 900 python tools/$S $A > $O/${S%.py}.txt 2>&1; echo "$S rc $?"; grep -v amdgpu.ids $O/${S%.py}.txt | tail -40 | cut -c1-220 ;;
    *)      echo "unknown step $STEP" ;;
  esac
done
