# round 5: same-box A/B of library builds / switches (fit + predict at C3), optional tests and timeline
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r05_ab}; mkdir -p $OUT
NEW=$GRAFT_REPO_ROOT/scikit-gpuppy_amd/skgpuppy_amd/libgpx.so
OLD=$GRAFT_REPO_ROOT/tools/native/libgpx_r04.so
if [ "$2" != "notest" ]; then
timeout -k 10 900 python -m pytest tests/test_dataflow.py tests/test_gpu_parity.py -m gpu -x -q -k "dataflow or bit_reproducible or fallback or stalled or jitter or reconstructs or c3_fit or potrf or trapezoid or full_size_properties or fit_golden or estimate_many_golden or c2_full or against_oracle_ragged" > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -5 $OUT/tests.log
fi
timeout -k 10 1500 python tools/probe_fit_lib.py $OLD $NEW ${AB_VARIANTS} 2>&1 | tee $OUT/ab.txt
PROBE_N=4096 PROBE_D=4 timeout -k 10 300 python tools/probe_fit_lib.py $OLD $NEW 2>&1 | tee $OUT/ab_c2.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-propagate --no-extras --no-cpu --no-python-api --warmup 1 --steps 3 > $GRAFT_REPO_ROOT/$OUT/trace_line.json 2> $GRAFT_REPO_ROOT/$OUT/trace.err
cd $GRAFT_REPO_ROOT
db=$(ls $OUT/trace/*/*.db $OUT/trace/*.db 2>/dev/null | head -1)
python tools/fit_timeline.py $db 2 detail > $OUT/timeline.txt 2>&1
python tools/fit_timeline.py $db 2 all 0 40000 > $OUT/timeline_all.txt 2>&1
rm -rf $OUT/trace
head -24 $OUT/timeline.txt
