#!/bin/bash
# lab: VGPR / AGPR / spill counts of every kernel of an assembly listing produced with  hipcc -S --cuda-device-only
python3 - "$1" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    print(f"{m.group(2)[:90]:90s} vgpr {m.group(3):>4s} agpr {m.group(1):>4s} spill {m.group(4)}")
PY
