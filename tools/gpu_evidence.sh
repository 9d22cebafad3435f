#!/bin/bash
# Round evidence on the GPU box: the bench line, the same command under rocprofv3 --kernel-trace --stats, the headline
# alone under the profiler, and three PMC passes of the headline (one counter group per run).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 500 python bench.py > $O/r2_bench_final.json 2> $O/r2_bench_final.err || { tail -5 $O/r2_bench_final.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r2_prof_final $O/r2_prof_headline $O/r2_pmc_rd $O/r2_pmc_wr $O/r2_pmc_sq
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/r2_prof_final -- python3 $B > $O/r2_bench_final_rocprof.json 2> $O/r2_bench_final_rocprof.err
echo "rocprof full rc=$?"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/r2_prof_headline -- python3 $B --no-extra --no-cpu-baseline > $O/r2_headline_rocprof.json 2> $O/r2_headline_rocprof.err
echo "rocprof headline rc=$?"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r2_pmc_rd -- python3 $B --no-extra --no-cpu-baseline > $O/r2_pmc_rd.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r2_pmc_wr -- python3 $B --no-extra --no-cpu-baseline > $O/r2_pmc_wr.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/r2_pmc_sq -- python3 $B --no-extra --no-cpu-baseline > $O/r2_pmc_sq.log 2>&1 || exit 1
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py gpurun_out/r2_pmc_hbm_traffic.json gpurun_out/r2_pmc_rd gpurun_out/r2_pmc_wr gpurun_out/r2_pmc_sq
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
ls gpurun_out/r2_prof_final/* gpurun_out/r2_prof_headline/*
