#!/bin/bash
# One PMC pass (SQ instruction mix, LDS bank conflicts) over the feat leg: what bounds k_feat_sliding_lds.
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/r2_pmcs
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/r2_pmcs -- python3 $GRAFT_REPO_ROOT/bench.py --legs feat --no-cpu-baseline --steps 1 --warmup 0 > $O/r2_pmcs.log 2>&1 || { tail -5 $O/r2_pmcs.log; exit 1; }
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r2_pmcs/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': n[k] += 1
for k in acc:
    if 'slid' in k or 'walk' in k:
        print(k, n[k], {c: round(v / max(1, n[k])) for c, v in acc[k].items()})
PY
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
