cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r05_host}; mkdir -p $OUT
GPX_DEBUG=1 PROBE_REPS=6 python tools/probe_fit_lib.py scikit-gpuppy_amd/skgpuppy_amd/libgpx.so > $OUT/out.txt 2> $OUT/err.txt
grep "host enqueue" $OUT/err.txt | tail -3
cat $OUT/out.txt
GPX_DEBUG=1 PROBE_REPS=6 PROBE_N=4096 PROBE_D=4 python tools/probe_fit_lib.py scikit-gpuppy_amd/skgpuppy_amd/libgpx.so > $OUT/out2.txt 2> $OUT/err2.txt
grep "host enqueue" $OUT/err2.txt | tail -3
cat $OUT/out2.txt
