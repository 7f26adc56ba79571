cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tools/probe_multi.py 16384 8 1 2
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_device_abi" 2>&1 | tail -2
