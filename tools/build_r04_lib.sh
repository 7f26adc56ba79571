#!/bin/bash
# Rebuild the round-4 library (commit c486b7f) as tools/native/libgpx_r04.so for same-box A/B runs (tools/probe_fit_lib.py,
# tools/sharded_ab.sh, tools/r05_*.sh).  Needs the git history; the .so is git-ignored and travels with gpurun.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive c486b7f scikit-gpuppy_amd include | tar -x -C "$TMP"
make -C "$TMP/scikit-gpuppy_amd/csrc" -j8 > /dev/null
cp "$TMP/scikit-gpuppy_amd/skgpuppy_amd/libgpx.so" "$ROOT/tools/native/libgpx_r04.so"
rm -rf "$TMP"
ls -la "$ROOT/tools/native/libgpx_r04.so"
