cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tools/probe_multi.py 16384 8 1 2 4
timeout -k 10 600 python3 tools/probe_multi.py 65536 16 1 2
