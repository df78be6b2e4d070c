#!/bin/bash
set -e
for st in 3 4 99; do
  FUSED_STAGE=$st python tools/gen_fused_asm.py > /dev/null
  make -C adalog_amd/csrc 2>&1 | grep -E "error:" && exit 1
  echo "== stage $st"
  timeout 60 python tools/bench_fused.py 1536 384 197 4 4 2>&1 | grep -E "fused=|fault|Error|error" | head -3 || true
done
