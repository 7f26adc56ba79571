#!/bin/bash
# Round-end measurement set on the GPU box (gpurun -- 'bash tools/run_round_profile.sh TAG'):
#   default bench line (with the CPU leg), kernel-stats CSV of the exact timed region, PMC traffic passes.
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-propagate --no-extras --no-cpu --no-python-api --warmup 0 --steps 5 > $OUT/stats_line.json 2> $OUT/stats.err; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-propagate --no-extras --no-python-api > $OUT/pmc_f.json 2> $OUT/pmc_f.err; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-propagate --no-extras --no-python-api > $OUT/pmc_w.json 2> $OUT/pmc_w.err; echo "pmc write rc=$?"
rocprofv3 --kernel-trace -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-propagate --no-extras --no-cpu --no-python-api --warmup 1 --steps 3 > $OUT/trace_line.json 2> $OUT/trace.err; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py $OUT/pmc_f $OUT/pmc_w $OUT/pmc_traffic.json
db=$(ls $OUT/trace/*/*.db $OUT/trace/*.db 2>/dev/null | head -1)
python tools/fit_timeline.py $db 2 detail > $OUT/cholesky_timeline.txt 2>&1
python tools/trace_db.py $db > $OUT/kernel_trace_summary.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_f $OUT/pmc_w
ls $OUT $OUT/stats | head -40
find $OUT/stats -name '*kernel_stats.csv' | head -2
