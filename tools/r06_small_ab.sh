cd $GRAFT_REPO_ROOT
for l in tools/native/libgpx_r05.so scikit-gpuppy_amd/skgpuppy_amd/libgpx.so tools/native/libgpx_r05.so scikit-gpuppy_amd/skgpuppy_amd/libgpx.so; do echo "== $l"; timeout -k 10 200 python3 tools/probe_small_gemm.py $l; done
