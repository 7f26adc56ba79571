cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r05_kinv}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kinv or nll or exact or propagation_golden or c3_fit" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
for v in "GPX_KINV_LAUUM=1" "GPX_KINV_LAUUM=0" "GPX_KINV_LAUUM=1" "GPX_KINV_LAUUM=0"; do echo $v; env $v python tools/probe_kinv.py 2>&1 | tail -2; done | tee $OUT/kinv.txt
