# round 5: the two-deep look-ahead schedule (chol.hip) -- tests of the factorisation, A/B of the bench, timeline
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r05_m1}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_dataflow.py tests/test_gpu_parity.py -m gpu -x -q -k "dataflow or bit_reproducible or fallback or stalled or jitter or reconstructs or c3_fit or potrf or trapezoid or full_size_properties or fit_golden" > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -5 $OUT/tests.log
B="python bench.py --no-cpu --no-python-api --no-propagate --no-extras"
for v in "X=1" "GPX_PANEL_INV=0" "GPX_RESERVE_CUS=0" "GPX_RESERVE_CUS=16" "GPX_RESERVE_CUS=24" "GPX_RESERVE_MIN_TILES=2000" "GPX_RESERVE_MIN_TILES=500" "GPX_SQK_TILES=1600"; do
  env $v timeout -k 10 120 $B --workload c3 --steps 10 --warmup 3 2> $OUT/c3_$v.err | tail -1 > $OUT/c3_$v.json
  python - "$v" $OUT/c3_$v.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read()); print("%-28s fit %.3f predict %.3f step %.3f value %.0f"%(sys.argv[1],d["fit_ms"],d["predict_ms"],d["ms_per_step"],d["value"]),flush=True)
except Exception as e: print(sys.argv[1],"FAILED",e,flush=True)
PY
done
for v in "X=1"; do
  env $v timeout -k 10 120 $B --workload c2 --steps 20 --warmup 5 2> $OUT/c2_$v.err | tail -1 > $OUT/c2_$v.json
  python - "$v" $OUT/c2_$v.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read()); print("c2 %-28s fit %.3f predict %.3f step %.3f value %.0f"%(sys.argv[1],d["fit_ms"],d["predict_ms"],d["ms_per_step"],d["value"]),flush=True)
except Exception as e: print(sys.argv[1],"FAILED",e,flush=True)
PY
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-propagate --no-extras --no-cpu --no-python-api --warmup 1 --steps 3 > $GRAFT_REPO_ROOT/$OUT/trace_line.json 2> $GRAFT_REPO_ROOT/$OUT/trace.err
cd $GRAFT_REPO_ROOT
db=$(ls $OUT/trace/*/*.db $OUT/trace/*.db 2>/dev/null | head -1)
python tools/fit_timeline.py $db 2 detail > $OUT/timeline.txt 2>&1
python tools/fit_timeline.py $db 2 all 0 40000 > $OUT/timeline_all.txt 2>&1
rm -rf $OUT/trace
head -30 $OUT/timeline.txt
