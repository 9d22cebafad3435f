#!/bin/bash
# GPU parity tests, every failure listed (no -x); args: test files (default: all of tests/)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
T="${*:-tests}"
timeout -k 10 900 python -u -m pytest $T -q -m gpu --timeout 180 --timeout-method thread -p no:cacheprovider > $O/r3_tests.log 2>&1
rc=$?
echo "pytest rc=$rc"
grep -E "^(FAILED|ERROR)|passed|failed" $O/r3_tests.log | cut -c1-200 | tail -70 || true
grep "AssertionError: (" $O/r3_tests.log | cut -c1-250 | tail -60 || true
exit $rc
