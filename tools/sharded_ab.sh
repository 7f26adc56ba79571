for v in "X=1" "GPX_TOP_EARLY=0" "GPX_SHARD_TIMING=0" "GPX_TOP_EARLY=0 GPX_SHARD_TIMING=0"; do
  env $v GPX_BENCH_SHARDED=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 10 200 python3 bench.py --gpus 1 --workload c3 --steps 4 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['fit_ms'], d['predict_ms'], d.get('per_rank_fit'))"
done
