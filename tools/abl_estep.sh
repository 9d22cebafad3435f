#!/bin/bash
# Timing of the ablation variants of the fat E-step kernel (tools/build_variant.sh) on one box: results are wrong by
# construction, only the kernel time is read.
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== full"; timeout -k 10 120 python tools/time_estep.py 512 2>&1 | grep "C=" || exit 1
for v in "$@"; do
  echo "== $v"; SKM_LIB=$GRAFT_REPO_ROOT/tools/ab/libskm_$v.so timeout -k 10 120 python tools/time_estep.py 512 2>&1 | grep "C=" || exit 1
done
echo "== full"; timeout -k 10 120 python tools/time_estep.py 512 2>&1 | grep "C=" || exit 1
