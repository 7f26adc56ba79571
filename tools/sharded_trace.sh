# kernel trace of the one-rank rehearsal of the sharded path at C3, panel message in two parts and in one: per-panel timeline of the last fit
# (tools/sharded_timeline.py) -> gpurun_out/sharded_timeline_{split,whole}.txt
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export GPX_BENCH_SHARDED=1 GPX_BENCH_SKIP_1GPU_REF=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
for msg in split whole; do
  export GPX_PANEL_MESSAGE=$msg
  rm -rf $ROOT/gpurun_out/shtrace_$msg
  timeout -k 10 400 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/shtrace_$msg -o t -- python3 $ROOT/bench.py --gpus 1 --workload c3 --steps 2 --warmup 1 > $ROOT/gpurun_out/shtrace_$msg.json 2> $ROOT/gpurun_out/shtrace_$msg.err || { tail -5 $ROOT/gpurun_out/shtrace_$msg.err; exit 1; }
  db=$(ls $ROOT/gpurun_out/shtrace_$msg/*.db $ROOT/gpurun_out/shtrace_$msg/*/*.db 2>/dev/null | head -1)
  python3 $ROOT/tools/sharded_timeline.py $db > $ROOT/gpurun_out/sharded_timeline_$msg.txt 2>&1
  cp $db $ROOT/gpurun_out/shtrace_$msg.db; rm -rf $ROOT/gpurun_out/shtrace_$msg
  head -40 $ROOT/gpurun_out/sharded_timeline_$msg.txt
done
