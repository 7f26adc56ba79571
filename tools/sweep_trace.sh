# kernel trace of the few-right-hand-side sweeps (propagate_GA right after a fit): launches behind the last ts_pack_kernel
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in ${1:-1 0}; do
  export GPX_SWEEP_LOOKAHEAD=$v
  rm -rf $ROOT/gpurun_out/swtrace
  timeout -k 10 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/swtrace -o t -- python3 $ROOT/bench.py --no-cpu --no-python-api --no-extras --steps 2 --warmup 1 > /dev/null 2> $ROOT/gpurun_out/swtrace.err || { tail -5 $ROOT/gpurun_out/swtrace.err; exit 1; }
  db=$(ls $ROOT/gpurun_out/swtrace/*.db $ROOT/gpurun_out/swtrace/*/*.db 2>/dev/null | head -1)
  echo "== GPX_SWEEP_LOOKAHEAD=$v"
  python3 $ROOT/tools/trace_list.py $db ts_pack -1 120 > $ROOT/gpurun_out/sweep_trace_$v.txt
  head -70 $ROOT/gpurun_out/sweep_trace_$v.txt
  rm -rf $ROOT/gpurun_out/swtrace
done
