cd $GRAFT_REPO_ROOT
L=scikit-gpuppy_amd/skgpuppy_amd/libgpx.so
for r in 1 2 3; do for v in 0 1; do echo "GPX_OCC3=$v: $(GPX_OCC3=$v timeout -k 10 200 python3 tools/probe_kinv.py 2>&1 | grep exact | tail -1)"; done; done
ROUNDS=4 PROBE_REPS=10 timeout -k 10 900 python3 tools/probe_fit_lib.py GPX_OCC3=0@$L GPX_OCC3=1@$L 2>&1 | tail -3
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or c3 or c2 or estimate_many or predict or kinv or exact or nll" 2>&1 | tail -3
