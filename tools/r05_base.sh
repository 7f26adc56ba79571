set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_base; mkdir -p $OUT
B="python bench.py --no-cpu --no-python-api --no-propagate --no-extras"
$B --workload c2 --steps 20 --warmup 5 > $OUT/c2_default.json 2> $OUT/c2_default.err
GPX_DFLOW_FROM=0 $B --workload c2 --steps 20 --warmup 5 > $OUT/c2_dflow.json 2> $OUT/c2_dflow.err
GPX_SQK_FROM=0 GPX_RESERVE_CUS=0 $B --workload c2 --steps 20 --warmup 5 > $OUT/c2_sqk_all.json 2> $OUT/c2_sqk_all.err
$B --workload c3 --steps 10 --warmup 3 > $OUT/c3_default.json 2> $OUT/c3_default.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_base/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "fit %.3f predict %.3f step %.3f value %.0f"%(d["fit_ms"],d["predict_ms"],d["ms_per_step"],d["value"]))
    except Exception as e: print(f,"FAILED",e)
PY
