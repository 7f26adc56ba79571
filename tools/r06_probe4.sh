cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/r06_probe3.sh
timeout -k 10 900 python3 -m pytest tests/test_dataflow.py tests/test_gpu_parity.py -x -q -m gpu -k "dataflow or golden or reproducible or kat1 or c3 or c2 or gram or stall" > gpurun_out/r06_pytest_subset.txt 2>&1; tail -3 gpurun_out/r06_pytest_subset.txt
PROBE_N=4096 PROBE_D=4 ROUNDS=3 timeout -k 10 400 python3 tools/probe_fit_lib.py tools/native/libgpx_r05.so scikit-gpuppy_amd/skgpuppy_amd/libgpx.so 2>&1 | tail -2
ROUNDS=2 timeout -k 10 400 python3 tools/probe_fit_lib.py tools/native/libgpx_r05.so scikit-gpuppy_amd/skgpuppy_amd/libgpx.so 2>&1 | tail -2
