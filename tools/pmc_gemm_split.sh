#!/bin/bash
# Counter passes over the split GEMM at the L-build shape (one counter group per run, the program directly after --).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
T=$GRAFT_REPO_ROOT/tools/time_gemm_split.py
export ONLY=lbuild
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_b3_*
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_b3_a -- python3 $T > $O/pmc_b3_a.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $O/pmc_b3_b -- python3 $T > $O/pmc_b3_b.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_b3_c -- python3 $T > $O/pmc_b3_c.log 2>&1 || exit 1
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, collections
for tag in "abc":
    agg = collections.defaultdict(lambda: [0.0, 0])
    for fn in glob.glob("gpurun_out/pmc_b3_%s/**/*counter_collection.csv" % tag, recursive=True):
        for row in csv.DictReader(open(fn)):
            if "k_gemm_b3" in row["Kernel_Name"]:
                a = agg[row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    for k, (v, n) in sorted(agg.items()):
        print("%-40s %16.0f per launch (%d launches)" % (k, v / n, n))
PY
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
