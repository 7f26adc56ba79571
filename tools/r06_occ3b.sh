cd $GRAFT_REPO_ROOT
echo "== two workgroups per CU (shipped kernel)"; GPX_OCC3=0 timeout -k 10 300 python3 tools/probe_gemm_k.py 2>&1 | grep gpx
echo "== three workgroups per CU (occ3)"; GPX_OCC3=2 timeout -k 10 300 python3 tools/probe_gemm_k.py 2>&1 | grep gpx
