#!/bin/bash
# Timing experiments on the hand-scheduled fused loop (results are WRONG in most variants; only the kernel time matters).
# Run on the GPU box from the repo root:  bash tools/lab/variants_fused.sh "<VAR=1 ...>" "<...>" ...
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "$@"; do
  env $v FUSED_NRB=12 FUSED_FNS=4 python3 tools/gen_fused_asm.py > /dev/null
  make -C adalog_amd/csrc 2>&1 | grep -E " error" | head -3
  rm -rf /tmp/vf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vf -o p -- python3 tools/bench_fused.py > /tmp/vf.log 2>&1
  echo "== variant: $v   $(grep 'fused=True' /tmp/vf.log)"
  python3 - <<PY
import csv
for r in csv.DictReader(open("/tmp/vf/p_kernel_stats.csv")):
    if "k_act_fused" in r["Name"] or "k_tie_flags" in r["Name"]:
        print("     ", round(float(r["AverageNs"]) / 1e3, 1), "us", r["Name"][:60])
PY
done
FUSED_NRB=12 FUSED_FNS=4 python3 tools/gen_fused_asm.py > /dev/null
make -C adalog_amd/csrc 2>&1 | grep -E " error" | head -3
