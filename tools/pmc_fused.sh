#!/bin/bash
# PMC passes over the fused activation-search kernel (deit_small fc2 shape; run on the GPU box from the repo root):
#   bash tools/pmc_fused.sh <out_dir>
# One rocprofv3 --pmc run per counter group (no trace domains alongside --pmc); prints per-dispatch averages of k_act_fused*.
set -u
out=${1:-gpurun_out/pmc_fused}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
groups=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum")
i=0
for g in "${groups[@]}"; do
  d="$out/g$i"
  rocprofv3 --pmc $g --output-format csv -d "$d" -o p -- python3 tools/bench_fused.py > "$d.log" 2>&1 < /dev/null
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, glob, json, os, sys
root = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_act_fused" not in row["Kernel_Name"]:
            continue
        a = acc.setdefault(row["Counter_Name"], [0.0, 0])
        a[0] += float(row["Counter_Value"]); a[1] += 1
res = {k: v[0] / max(v[1], 1) for k, v in sorted(acc.items())}
json.dump(res, open(os.path.join(root, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$out"/g*/
