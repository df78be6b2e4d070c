#!/usr/bin/env python3
"""Average the per-dispatch counter values of the scoring GEMM kernel over the rocprofv3 CSVs written by pmc_gemm.sh."""
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]
KERNEL = "k_gemm_"
out = {}
for d in sorted(glob.glob(os.path.join(root, "*_*_*"))):
    if not os.path.isdir(d):
        continue
    shape, mode, _ = os.path.basename(d).split("_", 2)
    dst = out.setdefault(f"{shape}_{mode}", {})
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                m = re.search(r"k_gemm_\w+<[^>]*>", row["Kernel_Name"])
                dst["kernel"] = m.group(0) if m else row["Kernel_Name"][:60]
        for k, v in acc.items():
            dst[k] = sum(v) / len(v)
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Name"]:
                dst["avg_us"] = float(row["AverageNs"]) / 1e3
for k, v in out.items():
    if "FETCH_SIZE" in v:                                   # KiB units; gfx950: FETCH_SIZE counts half of wide reads
        v["hbm_read_MB_corrected"] = 2.0 * v["FETCH_SIZE"] / 1024.0
    if "WRITE_SIZE" in v:
        v["hbm_write_MB"] = v["WRITE_SIZE"] / 1024.0
    if "GRBM_GUI_ACTIVE" in v and "avg_us" in v:
        v["clock_GHz"] = v["GRBM_GUI_ACTIVE"] / 8.0 / v["avg_us"] / 1e3
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:   # busy SIMD-cycles / (kernel cycles x 256 CUs x 4 SIMDs)
        v["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
    if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v:
        v["wait_any_frac"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in v:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
print(json.dumps(out, indent=1))
