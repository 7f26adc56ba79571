# one-rank rehearsal of the sharded path on the GPU box: fit / predict and the owner's panel-step times at C3 (and C4 with "c4")
# through the current library and, for comparison, the round-4 library (OLDLIB, via GPX_LIB)
cd $GRAFT_REPO_ROOT
for wl in ${1:-c3}; do
for lib in scikit-gpuppy_amd/skgpuppy_amd/libgpx.so ${OLDLIB}; do
  GPX_LIB=$GRAFT_REPO_ROOT/$lib GPX_BENCH_SHARDED=1 GPX_BENCH_SKIP_1GPU_REF=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 500 python3 bench.py --gpus 1 --workload $wl --steps ${STEPS:-4} --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl $lib', 'fit_ms', d['fit_ms'], 'predict_ms', d['predict_ms'], d.get('per_rank_fit'))"
done
done
