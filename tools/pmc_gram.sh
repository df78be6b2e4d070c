#!/bin/bash
# PMC passes over the Gram-form kernels as PRODUCTION dispatches them (run on the GPU box from the repo root):
#   bash tools/pmc_gram.sh <out_dir>
# k_gram_score (weight searches, csrc/gram.hip) and k_ga_quad (activation searches, csrc/gram_act.hip) at the deit_small qkv
# shape, with their build kernels.  One rocprofv3 --pmc run per counter group (no trace domains alongside --pmc), then per-kernel
# averages -> <out_dir>/summary.json.
set -u
out=${1:-gpurun_out/pmc_gram}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
groups=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS")
for prog in "gram_check.py prof" "gram_act_check.py x"; do
  tag=$(echo $prog | cut -d. -f1)
  i=0
  for g in "${groups[@]}"; do
    d="$out/${tag}_$i"
    rocprofv3 --pmc $g --output-format csv -d "$d" -o p -- python3 tools/lab/$prog > "$d.log" 2>&1 < /dev/null
    i=$((i+1))
  done
  d="$out/${tag}_trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o p -- python3 tools/lab/$prog > "$d.log" 2>&1 < /dev/null
done
python3 - "$out" <<'PY'
import csv, glob, json, os, re, sys
root = sys.argv[1]
out = {}
def key(name):
    m = re.search(r"(k_gram_\w+|k_ga_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for row in csv.DictReader(open(f)):
        k = key(row["Kernel_Name"])
        if k:
            acc.setdefault((k, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    for (k, c), v in acc.items():
        out.setdefault(k, {})[c] = sum(v) / len(v)
for f in glob.glob(os.path.join(root, "*", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = key(row["Name"])
        if k:
            out.setdefault(k, {})["avg_us"] = float(row["AverageNs"]) / 1e3
            out[k]["calls"] = int(row["Calls"])
for k, v in out.items():
    if "FETCH_SIZE" in v:                                   # KiB units; gfx950: FETCH_SIZE counts half of wide reads
        v["hbm_read_MB_corrected"] = 2.0 * v["FETCH_SIZE"] / 1024.0
    if "WRITE_SIZE" in v:
        v["hbm_write_MB"] = v["WRITE_SIZE"] / 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        v["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
    if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v:
        v["wait_any_frac"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
    if "SQ_ACTIVE_INST_VALU" in v and "SQ_WAVE_CYCLES" in v:
        v["valu_active_frac_of_wave_cycles"] = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -f "$out"/*/p_counter_collection.csv "$out"/*/*/p_counter_collection.csv
