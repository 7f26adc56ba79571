cd $GRAFT_REPO_ROOT
L=scikit-gpuppy_amd/skgpuppy_amd/libgpx.so
ROUNDS=3 PROBE_REPS=10 timeout -k 10 900 python3 tools/probe_fit_lib.py GPX_OCC3=0@$L GPX_OCC3=1@$L 2>&1 | tail -3
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or c3 or c2 or estimate_many or predict" 2>&1 | tail -3
