#!/bin/bash
# Timing experiments on the hand-scheduled fused loop (results are WRONG in most variants; only the kernel time matters).
# Run on the GPU box from the repo root:  bash tools/lab/variants_fused.sh "<VAR=1 ...>" "<...>" ...   ("base" = the shipped loop)
# Switches of tools/gen_fused_asm.py: FUSED_NOMFMA, FUSED_NOBAR, FUSED_NOCOLD, FUSED_NOAREAD (no A-fragment LDS reads), FUSED_NOLUT (no LUT
# reads), FUSED_NOTIE (no near-tie tracking), FUSED_NOGEN (no fragment generation), FUSED_NODMA (weight ring never refilled).
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
make -C adalog_amd/csrc -j16 2>&1 | grep -E " error" | head -3
for v in "$@"; do
  if [ "$v" = "base" ]; then vv="FUSED_BASE=1"; else vv="$v"; fi
  env $vv FUSED_NRB=12 FUSED_FNS=4 python3 tools/gen_fused_asm.py > /dev/null
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
