cd $GRAFT_REPO_ROOT
for v in default 0; do
  if [ $v = default ]; then unset GPX_SQK_FROM; else export GPX_SQK_FROM=$v; fi
  echo "== GPX_SQK_FROM=$v"
  timeout -k 10 400 python3 tools/probe_multi.py 16384 8 1 2 2>&1 | grep -v amdgpu
  timeout -k 10 600 python3 tools/probe_multi.py 65536 16 1 2>&1 | grep -v amdgpu
done
