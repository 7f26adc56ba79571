cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_operator_interface.py -x -q -m gpu -k "spgp or operator or exact or golden" 2>&1 | tail -5
