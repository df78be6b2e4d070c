#!/bin/bash
# Timing experiments on the hand-scheduled fused loop (results are WRONG in most variants; only the kernel time matters).
# Run on the GPU box from the repo root:  bash tools/lab/variants_fused.sh "<VAR=1 ...>" "<...>" ...
set -u
for v in "$@"; do
  env $v FUSED_NRB=12 FUSED_FNS=4 python3 tools/gen_fused_asm.py > /dev/null
  make -C adalog_amd/csrc 2>&1 | grep -E " error" | head -3
  echo "== variant: $v"
  timeout 120 python3 tools/bench_fused.py 2>&1 | grep "fused=True"
done
FUSED_NRB=12 FUSED_FNS=4 python3 tools/gen_fused_asm.py > /dev/null
make -C adalog_amd/csrc 2>&1 | grep -E " error" | head -3
