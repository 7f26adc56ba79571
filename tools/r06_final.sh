cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r06_pytest_full.txt 2>&1; tail -4 gpurun_out/r06_pytest_full.txt
timeout -k 10 600 python3 bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r06_bench_final.json
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
