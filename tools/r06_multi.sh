cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "multi_device_abi" > gpurun_out/r06_multi_test.txt 2>&1; tail -30 gpurun_out/r06_multi_test.txt
