#!/bin/bash
# Feature post-processing on the GPU box: parity tests, the bench leg, two PMC passes of the sliding normalisation.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_feat.py -x -q -m gpu --timeout 120 --timeout-method thread > $O/feat_tests.log 2>&1 || { tail -20 $O/feat_tests.log; exit 1; }
tail -2 $O/feat_tests.log
timeout -k 10 200 python bench.py --legs feat --no-cpu-baseline > $O/bench_feat1.json 2> $O/bench_feat.err || exit 1
python -c "import json; d=json.load(open('$O/bench_feat1.json'))['feat']; print(d['value'], d['cmvn_sliding'])"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r2_pmcf_*
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r2_pmcf_rd1 -- python3 $GRAFT_REPO_ROOT/bench.py --legs feat --no-cpu-baseline --steps 1 --warmup 0 > $O/r2_pmcf.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r2_pmcf_wr1 -- python3 $GRAFT_REPO_ROOT/bench.py --legs feat --no-cpu-baseline --steps 1 --warmup 0 >> $O/r2_pmcf.log 2>&1 || exit 1
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py gpurun_out/r2_pmc_feat.json gpurun_out/r2_pmcf_rd1 gpurun_out/r2_pmcf_wr1 | grep -i "slid\|walk"
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
