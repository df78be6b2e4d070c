#!/bin/bash
# PMC passes over the scoring GEMM at the deit_small qkv / fc2 / q.k^T search shapes (slab, streaming and group kernels) (run on the GPU box from the repo root):
#   bash tools/pmc_gemm.sh <out_dir> ["qkv fc2 qk"]
# One rocprofv3 --pmc run per counter group (the TCC byte counters do not fit one pass; no trace domains alongside --pmc),
# then tools/pmc_summary.py averages the per-dispatch values of the GEMM kernel.
set -u
out=${1:-gpurun_out/pmc_stream}
shapes=${2:-"qkv fc2 qk"}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
groups=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
        "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC")
for shape in $shapes; do
  for mode in w a g; do
    if [ "$shape" = qk ] && [ "$mode" != a ]; then continue; fi
    if [ "$shape" = fc2 ] && [ "$mode" = g ]; then continue; fi
    i=0
    for g in "${groups[@]}"; do
      d="$out/${shape}_${mode}_$i"
      rocprofv3 --pmc $g --output-format csv -d "$d" -o p -- python3 tools/prof_gemm.py $shape $mode > "$d.log" 2>&1 < /dev/null
      i=$((i+1))
    done
    d="$out/${shape}_${mode}_trace"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o p -- python3 tools/prof_gemm.py $shape $mode > "$d.log" 2>&1 < /dev/null
  done
done
python3 tools/pmc_summary.py "$out" > "$out/summary.json"
cat "$out/summary.json"
