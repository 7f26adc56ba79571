set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r05_shard}; mkdir -p $OUT
timeout -k 10 1200 python -m pytest tests/test_gpu_parity.py tests/test_dataflow.py -m gpu -x -q -k "sharded or panel or two_threads or fallback or stalled or reproducible or dataflow or spgp" > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -8 $OUT/tests.log
OLDLIB=tools/native/libgpx_r04.so bash tools/sharded_ab.sh c3 2>&1 | tee $OUT/c3.txt
OLDLIB=tools/native/libgpx_r04.so STEPS=2 bash tools/sharded_ab.sh c4 2>&1 | tee $OUT/c4.txt
