cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 240 tools/native/probe_tile_stamps.bin 1024 2048 > gpurun_out/r06_tile_stamps_new.txt 2>&1 || { tail -5 gpurun_out/r06_tile_stamps_new.txt; exit 1; }
grep -A12 "beta=1" gpurun_out/r06_tile_stamps_new.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or reproducible or kat1 or c3 or gram" > gpurun_out/r06_pytest_subset.txt 2>&1; tail -3 gpurun_out/r06_pytest_subset.txt
ROUNDS=2 timeout -k 10 400 python3 tools/probe_fit_lib.py tools/native/libgpx_r05.so scikit-gpuppy_amd/skgpuppy_amd/libgpx.so 2>&1 | tail -4
PROBE_N=4096 PROBE_D=4 ROUNDS=2 timeout -k 10 400 python3 tools/probe_fit_lib.py tools/native/libgpx_r05.so scikit-gpuppy_amd/skgpuppy_amd/libgpx.so 2>&1 | tail -4
