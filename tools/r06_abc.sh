cd $GRAFT_REPO_ROOT
ROUNDS=5 PROBE_REPS=10 timeout -k 10 1000 python3 tools/probe_fit_lib.py tools/native/libgpx_r05.so tools/native/libgpx_E.so tools/native/libgpx_D.so 2>&1 | tail -4
PROBE_N=4096 PROBE_D=4 ROUNDS=3 PROBE_REPS=12 timeout -k 10 400 python3 tools/probe_fit_lib.py tools/native/libgpx_r05.so tools/native/libgpx_E.so 2>&1 | tail -3
