cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/sqk_trace.*
GPX_DFLOW_TRACE=gpurun_out/sqk_trace PROBE_N=4096 PROBE_D=4 PROBE_REPS=5 timeout -k 10 240 python3 tools/probe_fit_lib.py scikit-gpuppy_amd/skgpuppy_amd/libgpx.so > gpurun_out/r06_sqk_trace_run.txt 2>&1 || { tail -5 gpurun_out/r06_sqk_trace_run.txt; exit 1; }
for i in 16 17 19; do echo "== square launch $i"; python3 tools/sqk_steps.py gpurun_out/sqk_trace.$i; done > gpurun_out/r06_sqk_steps.txt 2>&1
grep -E "^==|^mean|rows below" gpurun_out/r06_sqk_steps.txt
rm -f gpurun_out/sqk_trace.*
