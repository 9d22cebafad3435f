"""Summarise rocprofv3 --pmc passes (one directory per pass, CSV output) into profiles/<name>.json:
per kernel, the average counter values per launch, HBM bytes per launch (FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950) and the MFMA utilisation
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs).
Usage: python tools/pmc_summary.py OUT.json PASSDIR [PASSDIR ...]"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = {}
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if not k.startswith("k_"):
                    continue
                key = (k, row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
            for (k, _, c), v in per_dispatch.items():
                a = acc.setdefault(k, {}).setdefault(c, [0.0, 0])
                a[0] += v
                a[1] += 1
    kernels = {}
    for k, cs in sorted(acc.items()):
        rec = {c: v[0] / v[1] for c, v in cs.items()}
        rec["launches"] = max(v[1] for v in cs.values())
        if "FETCH_SIZE" in rec:
            rec["hbm_read_bytes_per_launch"] = rec["FETCH_SIZE"] * 1024.0 * 2.0
        if "WRITE_SIZE" in rec:
            rec["hbm_write_bytes_per_launch"] = rec["WRITE_SIZE"] * 1024.0
        if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
            rec["hbm_bytes_per_launch"] = rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"]
        if rec.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in rec:
            rec["mfma_util"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (rec["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        kernels[k] = rec
    json.dump({"kernels": kernels}, open(out, "w"), indent=1)
    for k, r in kernels.items():
        print(k, r.get("launches"), "mfma_util=%.3f" % r.get("mfma_util", float("nan")),
              "hbm_MB=%.1f" % (r.get("hbm_bytes_per_launch", float("nan")) / 1e6))


if __name__ == "__main__":
    main()
