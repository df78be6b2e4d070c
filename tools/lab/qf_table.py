"""Lab: per-kernel time of the last N quant_forward passes of a rocprofv3 kernel trace of tools/lab/qf_prof.py."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
names = [r["Kernel_Name"] for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
mark = [i for i, n in enumerate(names) if "triu_tril" in n]      # qf_prof.py launches it right before the measured passes
lo = mark[-1] + 1 if mark else 0
agg, tot = collections.OrderedDict(), 0.0
BY_GRID = len(sys.argv) > 3
grids = [r.get("Grid_Size", r.get("Grid_Size_X", "?")) for r in rows]
for i, (n, d) in enumerate(zip(names[lo:], dur[lo:])):
    k = n.replace("(anonymous namespace)::", "").replace("at::native::", "").split("(")[0][:100]
    if BY_GRID and "k_gemm_cand" in k:
        k += f" grid={grids[lo + i]}"
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d; tot += d
print("markers", len(mark), "GPU us per forward", round(tot / N, 1))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{v[1] / N:9.1f} us/fwd {v[0] / N:6.1f} calls  {k}")
