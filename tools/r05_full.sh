# round 5: the whole GPU suite, then a same-box A/B against the round-4 library, the SPGP C5 figures and the one-rank sharded rehearsal
set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r05_full}; mkdir -p $OUT
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -15 $OUT/tests.log
NEW=$GRAFT_REPO_ROOT/scikit-gpuppy_amd/skgpuppy_amd/libgpx.so
OLD=$GRAFT_REPO_ROOT/tools/native/libgpx_r04.so
timeout -k 10 600 python tools/probe_fit_lib.py $OLD $NEW ${AB_VARIANTS} 2>&1 | tee $OUT/ab.txt
timeout -k 10 300 python tools/bench_spgp.py 2>&1 | tail -3 | tee $OUT/spgp.txt
GPX_LIB=$OLD timeout -k 10 300 python tools/bench_spgp.py 2>&1 | tail -3 | tee $OUT/spgp_r04lib.txt
