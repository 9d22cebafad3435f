#!/bin/bash
# Same-box A/B of two builds of libskm.so on the E-step: tools/ab/libskm_base.so (a copy of an earlier build, not tracked) against the in-tree library.
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
  echo "== base";  SKM_LIB=$GRAFT_REPO_ROOT/tools/ab/libskm_base.so timeout -k 10 120 python tools/time_estep.py "$@" 2>&1 | grep "C=" || exit 1
  echo "== new";   timeout -k 10 120 python tools/time_estep.py "$@" 2>&1 | grep "C=" || exit 1
done
