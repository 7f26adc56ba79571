cd $GRAFT_REPO_ROOT
L=scikit-gpuppy_amd/skgpuppy_amd/libgpx.so
ROUNDS=4 PROBE_REPS=10 timeout -k 10 900 python3 tools/probe_fit_lib.py GPX_SQK_LEFT_ROWS=1000000@$L GPX_SQK_LEFT_ROWS=32@$L GPX_SQK_LEFT_ROWS=64@$L GPX_SQK_LEFT_ROWS=0@$L 2>&1 | tail -5
PROBE_N=4096 PROBE_D=4 ROUNDS=3 PROBE_REPS=12 timeout -k 10 400 python3 tools/probe_fit_lib.py GPX_SQK_LEFT_ROWS=1000000@$L GPX_SQK_LEFT_ROWS=16@$L GPX_SQK_LEFT_ROWS=0@$L 2>&1 | tail -4
PROBE_N=8192 PROBE_D=8 ROUNDS=3 PROBE_REPS=12 timeout -k 10 400 python3 tools/probe_fit_lib.py GPX_SQK_LEFT_ROWS=1000000@$L GPX_SQK_LEFT_ROWS=32@$L GPX_SQK_LEFT_ROWS=0@$L 2>&1 | tail -4
