#!/bin/bash
# HBM traffic of the scoring GEMM over one whole calibration step (run on the GPU box from the repo root):
#   bash tools/pmc_bench.sh <out_dir>
# Two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains next to --pmc), then
# the per-launch averages over every dispatch of each scoring-GEMM instantiation -> <out_dir>/traffic.json.
set -u
out=${1:-gpurun_out/pmc_bench}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/$c" -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rerun-all > "$out/$c.log" 2>&1 < /dev/null
done
python3 - "$out" <<'PY'
import csv, glob, json, os, sys
root = sys.argv[1]
acc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if ("k_gemm_" not in name and "k_act_fused" not in name and "k_gram_" not in name and "k_ga_" not in name) or row["Counter_Name"] != c:
                continue
            key = name.split("(anonymous namespace)::")[1].split("(")[0]
            a = acc.setdefault(key, {}).setdefault(c, [0.0, 0])
            a[0] += float(row["Counter_Value"]); a[1] += 1
res = {}
for k, v in acc.items():
    n = v.get("FETCH_SIZE", [0, 0])[1]
    res[k] = {"launches": n,
              # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts half of wide (16 B/lane) reads: doubled (MI355X guide, HBM)
              "hbm_read_bytes_per_launch": 2.0 * 1024.0 * v["FETCH_SIZE"][0] / max(n, 1) if "FETCH_SIZE" in v else None,
              "hbm_write_bytes_per_launch": 1024.0 * v["WRITE_SIZE"][0] / max(v["WRITE_SIZE"][1], 1) if "WRITE_SIZE" in v else None}
json.dump(res, open(os.path.join(root, "traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -f "$out"/*/p_counter_collection.csv "$out"/*/*/p_counter_collection.csv
