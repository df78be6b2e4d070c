#!/bin/bash
# regenerates every hand-scheduled variant of the fused activation-search loop that gemm_fused.hip includes
cd "$(dirname "$0")/.."
for v in "12 4" "12 3" "8 4" "4 4"; do
  set -- $v
  FUSED_NRB=$1 FUSED_FNS=$2 python3 tools/gen_fused_asm.py
done
