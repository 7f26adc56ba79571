#!/bin/bash
# A/B of the factorisation schedule on the GPU box: fit_ms / predict_ms of the C3 bench for several environment settings,
# optionally followed by a kernel-trace timeline of one of them.   usage: tools/fit_ab.sh OUTTAG "ENV1" "ENV2" ... [-- "TIMELINE ENV"]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
    envs="$1"; shift
    line=$(env $envs python bench.py --no-cpu --no-python-api --no-propagate --no-extras --steps 10 --warmup 3 2> $OUT/ab_$i.err | tail -1)
    echo "$line" > $OUT/ab_$i.json
    python - "$envs" "$OUT/ab_$i.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read())
    print("%-60s fit %.2f ms  predict %.2f ms  step %.2f ms  value %.0f" % (sys.argv[1], d["fit_ms"], d["predict_ms"], d["ms_per_step"], d["value"]), flush=True)
except Exception as e:
    print("%-60s FAILED %s" % (sys.argv[1], e), flush=True)
PY
    i=$((i+1))
done
if [ "$1" == "--" ]; then
    shift
    envs="$1"
    cd /tmp && export TMPDIR=/tmp
    export $envs
    rocprofv3 --kernel-trace -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-propagate --no-extras --no-cpu --no-python-api --warmup 1 --steps 3 > $OUT/trace_line.json 2> $OUT/trace.err
    cd $GRAFT_REPO_ROOT
    db=$(ls $OUT/trace/*/*.db $OUT/trace/*.db 2>/dev/null | head -1)
    python tools/fit_timeline.py $db 2 detail > $OUT/timeline.txt 2>&1
    cat $OUT/timeline.txt | head -40
    python tools/fit_timeline.py $db 2 all ${TL_A:-18000} ${TL_B:-20000} > $OUT/timeline_all_mid.txt 2>&1
    python tools/fit_timeline.py $db 2 all ${TL_C:-24000} 40000 > $OUT/timeline_all_tail.txt 2>&1
    rm -rf $OUT/trace
fi
