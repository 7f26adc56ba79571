# round 6, first measurement call: (1) in-tile cycle stamps of the GEMM body, (2) step-by-step critical path of the square launches at C2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 240 tools/native/probe_tile_stamps.bin 1024 2048 > gpurun_out/r06_tile_stamps.txt 2>&1 || { tail -5 gpurun_out/r06_tile_stamps.txt; exit 1; }
cat gpurun_out/r06_tile_stamps.txt
rm -f gpurun_out/sqk_trace.*
GPX_DFLOW_TRACE=gpurun_out/sqk_trace PROBE_N=4096 PROBE_D=4 PROBE_REPS=5 timeout -k 10 240 python3 tools/probe_fit_lib.py scikit-gpuppy_amd/skgpuppy_amd/libgpx.so > gpurun_out/r06_sqk_trace_run.txt 2>&1 || { tail -5 gpurun_out/r06_sqk_trace_run.txt; exit 1; }
ls gpurun_out/sqk_trace.* | head -20
for i in 16 17 18 19; do echo "== square launch $i"; python3 tools/sqk_steps.py gpurun_out/sqk_trace.$i; done > gpurun_out/r06_sqk_steps.txt 2>&1
cat gpurun_out/r06_sqk_steps.txt
PROBE_N=4096 PROBE_D=4 PROBE_REPS=14 timeout -k 10 240 python3 tools/probe_fit_lib.py scikit-gpuppy_amd/skgpuppy_amd/libgpx.so
# parity of the rebuilt tile body (NEG MFMA, SGPR-addressed C) + a quick bench line + same-box A/B against the round-5 library
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or reproducible or kat1 or c3 or gram" > gpurun_out/r06_pytest_subset.txt 2>&1; tail -5 gpurun_out/r06_pytest_subset.txt
timeout -k 10 300 python3 bench.py --no-cpu --no-extras --no-python-api --no-propagate --steps 10 --warmup 3 > gpurun_out/r06_bench_quick.json 2> gpurun_out/r06_bench_quick.err; cut -c1-900 gpurun_out/r06_bench_quick.json
ROUNDS=2 timeout -k 10 400 python3 tools/probe_fit_lib.py tools/native/libgpx_r05.so scikit-gpuppy_amd/skgpuppy_amd/libgpx.so 2>&1 | tail -8
