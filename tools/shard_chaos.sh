# the 4-rank one-GPU rehearsal of the sharded fit (gloo with device tensors, C3 size) under GPX_SHARD_CHAOS with several seeds, split and whole message
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=2
for seed in ${SEEDS:-11 12 13 14 15 16}; do
  for msg in split whole; do
    GPX_SHARD_CHAOS=$seed GPX_PANEL_MESSAGE=$msg timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29737 tests/_gpu_shard_worker.py 16384 8 light gloo-device > gpurun_out/chaos_${seed}_${msg}.log 2>&1
    rc=$?
    echo "seed $seed message $msg rc=$rc $(grep -h 'sharded vs single-GPU' gpurun_out/chaos_${seed}_${msg}.log | head -1)"
    if [ $rc -ne 0 ]; then tail -20 gpurun_out/chaos_${seed}_${msg}.log; exit 1; fi
  done
done
