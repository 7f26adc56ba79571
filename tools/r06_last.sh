cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_device_abi" 2>&1 | tail -2
timeout -k 10 900 python3 tools/fuzz_parity.py 17 200 > gpurun_out/r06_fuzz_parity.txt 2>&1; tail -6 gpurun_out/r06_fuzz_parity.txt
