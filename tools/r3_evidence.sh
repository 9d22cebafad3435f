#!/bin/bash
# Round-3 evidence on the GPU box (tag = $1, default "final"): the bench line; the same command under
# rocprofv3 --kernel-trace --stats; the headline alone under the profiler; three PMC passes of the headline and three of
# the total-variability leg (one counter group per run, the program directly after --).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
TAG="${1:-final}"
O=$GRAFT_REPO_ROOT/gpurun_out
B=$GRAFT_REPO_ROOT/bench.py
P=r3_${TAG}
timeout -k 10 700 python bench.py > $O/${P}_bench.json 2> $O/${P}_bench.err || { tail -5 $O/${P}_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${P}_prof $O/${P}_prof_headline $O/${P}_pmc_rd $O/${P}_pmc_wr $O/${P}_pmc_sq $O/${P}_pmc_tv_rd $O/${P}_pmc_tv_wr $O/${P}_pmc_tv_sq $O/${P}_prof_tv
timeout -k 10 700 rocprofv3 --kernel-trace --stats -d $O/${P}_prof -- python3 $B > $O/${P}_bench_rocprof.json 2> $O/${P}_bench_rocprof.err
echo "rocprof full rc=$?"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/${P}_prof_headline -- python3 $B --no-extra --no-cpu-baseline > $O/${P}_headline_rocprof.json 2> $O/${P}_headline_rocprof.err
echo "rocprof headline rc=$?"
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${P}_pmc_rd -- python3 $B --no-extra --no-cpu-baseline > $O/${P}_pmc_rd.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${P}_pmc_wr -- python3 $B --no-extra --no-cpu-baseline > $O/${P}_pmc_wr.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${P}_pmc_sq -- python3 $B --no-extra --no-cpu-baseline > $O/${P}_pmc_sq.log 2>&1 || exit 1
TV="--legs tv --no-cpu-baseline --steps 1 --warmup 0"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/${P}_prof_tv -- python3 $B $TV > $O/${P}_prof_tv.log 2>&1
echo "tv trace rc=$?"
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${P}_pmc_tv_rd -- python3 $B $TV > $O/${P}_pmc_tv_rd.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${P}_pmc_tv_wr -- python3 $B $TV > $O/${P}_pmc_tv_wr.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/${P}_pmc_tv_sq -- python3 $B $TV > $O/${P}_pmc_tv_sq.log 2>&1 || exit 1
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py gpurun_out/${P}_pmc_hbm_traffic.json gpurun_out/${P}_pmc_rd gpurun_out/${P}_pmc_wr gpurun_out/${P}_pmc_sq
python - <<PY
import json
p = "gpurun_out/${P}_pmc_hbm_traffic.json"
d = json.load(open(p))
d = {"workload": {"C": 512, "D": 60, "frames": 10000000},
     "command": "bash tools/r3_evidence.sh ${TAG}: rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE} "
                "--output-format csv -- python3 bench.py --no-extra --no-cpu-baseline (one pass per run), summarised by tools/pmc_summary.py",
     **d}
json.dump(d, open(p, "w"), indent=1)
PY
python tools/pmc_summary.py gpurun_out/${P}_pmc_tv.json gpurun_out/${P}_pmc_tv_rd gpurun_out/${P}_pmc_tv_wr gpurun_out/${P}_pmc_tv_sq
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
# the trace databases are tens of MB each (gpurun merges at most 64 MiB back): keep their kernel statistics only
for d in prof prof_headline prof_tv; do
    db=$(ls gpurun_out/${P}_${d}/*/*_results.db | head -n 1)
    python tools/rocpd_stats.py "$db" gpurun_out/${P}_${d}_kernel_stats.csv
done
find gpurun_out -name "*_results.db" -delete
cat $O/${P}_bench.json
