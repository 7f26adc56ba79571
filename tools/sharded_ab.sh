# one-rank rehearsal of the sharded path on the GPU box: fit / predict, the owner's chain times and the exposed waits at C3 (and C4 with
# "c4"), the panel message split in head + tail (default) against one message per panel (GPX_PANEL_MESSAGE=whole), alternating
cd $GRAFT_REPO_ROOT
for wl in ${1:-c3}; do
for rnd in 1 2; do
for msg in split whole; do
  GPX_PANEL_MESSAGE=$msg GPX_BENCH_SHARDED=1 GPX_BENCH_SKIP_1GPU_REF=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 500 python3 bench.py --gpus 1 --workload $wl --steps ${STEPS:-4} --warmup 1 2>gpurun_out/sharded_ab.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl message=$msg', 'fit_ms %.2f' % d['fit_ms'], 'predict_ms %.2f' % d['predict_ms'], {k: round(v, 2) for k, v in d.get('per_rank_fit')[0].items()})" || tail -5 gpurun_out/sharded_ab.err
done
done
done
