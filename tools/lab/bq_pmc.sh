#!/bin/bash
# PMC passes over one case of tools/lab/bq_lab (run on the GPU box from the repo root):  bash tools/lab/bq_pmc.sh <out_dir> <case> [binary]
set -u
out=${1:-gpurun_out/bq_pmc}; cs=${2:-0}; bin=${3:-tools/lab/bq_lab}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
groups=("GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
        "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAVES SQ_INSTS_VMEM_RD")
i=0
for g in "${groups[@]}"; do
  rocprofv3 --pmc $g --output-format csv -d "$out/c${cs}_$i" -o p -- $bin $cs 5 > "$out/c${cs}_$i.log" 2>&1 < /dev/null
  i=$((i+1))
done
python3 - "$out" "$cs" <<'PY'
import csv, glob, sys, collections, json
out, cs = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/c{cs}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_bq_gemm" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in acc.items()}
json.dump(res, open(f"{out}/c{cs}_summary.json", "w"), indent=1)
for k in sorted(res): print(f"{k:28s} {res[k]:.4g}")
PY
