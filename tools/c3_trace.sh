# kernel trace of a C3 fit (N = 16384): every launch of the last-but-one fit: start / duration / idle-before per queue, and the panel timeline
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/c3trace
timeout -k 10 400 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/c3trace -o t -- python3 $ROOT/bench.py --no-cpu --no-python-api --no-extras --no-propagate --steps 3 --warmup 1 > $ROOT/gpurun_out/c3trace.json 2> $ROOT/gpurun_out/c3trace.err || { tail -5 $ROOT/gpurun_out/c3trace.err; exit 1; }
db=$(ls $ROOT/gpurun_out/c3trace/*.db $ROOT/gpurun_out/c3trace/*/*.db 2>/dev/null | head -1)
python3 $ROOT/tools/trace_list.py $db ts_pack 2 900 > $ROOT/gpurun_out/c3_trace.txt
python3 $ROOT/tools/fit_timeline.py $db 2 detail > $ROOT/gpurun_out/c3_timeline.txt 2>&1
rm -rf $ROOT/gpurun_out/c3trace
head -30 $ROOT/gpurun_out/c3_timeline.txt
