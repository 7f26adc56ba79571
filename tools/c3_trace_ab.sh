# same-box kernel traces of a C3 fit with the round-5 library and with the current one: per-panel timeline + per-kernel-class totals
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for tag in r05 new; do
  rm -rf $ROOT/gpurun_out/c3trace
  if [ $tag = r05 ]; then export GPX_LIB=$ROOT/tools/native/libgpx_r05.so; else unset GPX_LIB; fi
  timeout -k 10 400 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/c3trace -o t -- python3 $ROOT/bench.py --no-cpu --no-python-api --no-extras --no-propagate --steps 3 --warmup 1 > $ROOT/gpurun_out/c3trace_$tag.json 2> $ROOT/gpurun_out/c3trace.err || { tail -5 $ROOT/gpurun_out/c3trace.err; exit 1; }
  db=$(ls $ROOT/gpurun_out/c3trace/*.db $ROOT/gpurun_out/c3trace/*/*.db 2>/dev/null | head -1)
  python3 $ROOT/tools/trace_list.py $db ts_pack 2 900 > $ROOT/gpurun_out/c3_trace_$tag.txt
  python3 $ROOT/tools/fit_timeline.py $db 2 detail > $ROOT/gpurun_out/c3_timeline_$tag.txt 2>&1
  rm -rf $ROOT/gpurun_out/c3trace
  echo "== $tag"; head -22 $ROOT/gpurun_out/c3_timeline_$tag.txt
done
