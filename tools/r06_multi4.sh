cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tools/probe_multi.py 16384 8 1 2 4 2>&1 | grep -v amdgpu
timeout -k 10 1100 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_device_abi or sharded" 2>&1 | tail -3
