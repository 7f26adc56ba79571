# kernel trace of the C2 fit (N = 4096): every launch of the last fit, start / duration / idle-before per queue
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/c2trace
timeout -k 10 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/c2trace -o t -- python3 $ROOT/bench.py --workload c2 --no-cpu --no-python-api --no-extras --no-propagate --steps 3 --warmup 2 > $ROOT/gpurun_out/c2trace.json 2> $ROOT/gpurun_out/c2trace.err || { tail -5 $ROOT/gpurun_out/c2trace.err; exit 1; }
db=$(ls $ROOT/gpurun_out/c2trace/*.db $ROOT/gpurun_out/c2trace/*/*.db 2>/dev/null | head -1)
python3 $ROOT/tools/trace_list.py $db ts_pack -2 140 > $ROOT/gpurun_out/c2_trace.txt
python3 $ROOT/tools/fit_timeline.py $db 2 all 0 4000 > $ROOT/gpurun_out/c2_timeline.txt 2>&1
rm -rf $ROOT/gpurun_out/c2trace
head -150 $ROOT/gpurun_out/c2_timeline.txt
tail -1 $ROOT/gpurun_out/c2trace.json | cut -c1-300
