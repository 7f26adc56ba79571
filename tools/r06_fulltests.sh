cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -s -m gpu -k two_threads_large_fits 2>&1 | grep -E "two threads|passed|failed"; done > gpurun_out/r06_concurrent_ratio.txt 2>&1
cat gpurun_out/r06_concurrent_ratio.txt
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r06_pytest_full.txt 2>&1; tail -5 gpurun_out/r06_pytest_full.txt
