#!/bin/bash
# Round 3, first GPU call: the tightened parity tests (every failure listed, no -x), then PMC evidence for the TV leg
# (one counter group per run, the program directly after --).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 420 python -u -m pytest tests/test_gpu_raw_features.py tests/test_gpu_gmm.py tests/test_gpu_random_shapes.py tests/test_gpu_tv.py tests/test_zz_no_alloc.py -q -m gpu --timeout 120 --timeout-method thread -p no:cacheprovider > $O/r3_tests1.log 2>&1
echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $O/r3_tests1.log | cut -c1-300 | tail -60
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_pmc_tv_rd $O/r3_pmc_tv_wr $O/r3_pmc_tv_sq $O/r3_prof_tv
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/r3_prof_tv -- python3 $B --legs tv --no-cpu-baseline --steps 1 --warmup 0 > $O/r3_prof_tv.log 2>&1
echo "trace rc=$?"
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r3_pmc_tv_rd -- python3 $B --legs tv --no-cpu-baseline --steps 1 --warmup 0 > $O/r3_pmc_tv_rd.log 2>&1
echo "rd rc=$?"
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r3_pmc_tv_wr -- python3 $B --legs tv --no-cpu-baseline --steps 1 --warmup 0 > $O/r3_pmc_tv_wr.log 2>&1
echo "wr rc=$?"
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/r3_pmc_tv_sq -- python3 $B --legs tv --no-cpu-baseline --steps 1 --warmup 0 > $O/r3_pmc_tv_sq.log 2>&1
echo "sq rc=$?"
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py gpurun_out/r3_pmc_tv_start.json gpurun_out/r3_pmc_tv_rd gpurun_out/r3_pmc_tv_wr gpurun_out/r3_pmc_tv_sq
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
tail -3 $O/r3_prof_tv.log
