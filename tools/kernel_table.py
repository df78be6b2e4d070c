#!/usr/bin/env python3
"""Prints the per-kernel table of DESIGN.md section 7 from the two rocprofv3 kernel-stats CSVs of a round
(product schedule and reference schedule; three calibrations each)."""
import csv
import sys

r = sys.argv[1] if len(sys.argv) > 1 else "r06"
GROUPS = [
    ("`k_act_fused_asm` (fc2 activation search)", ["k_act_fused_asm"]),
    ("`k_ga_quad` + build + finish (Gram form: qkv / proj / fc1 activation searches)", ["k_ga_"]),
    ("`k_gram_score` + build (Gram form: qkv / proj / fc1 weight searches)", ["k_gram_"]),
    ("`k_i8mm` (the int8 product of both Gram builds)", ["k_i8mm"]),
    ("`k_gemm_slab<…, GEN>` (token-form activation searches)", ["k_gemm_slab<2, true", "k_gemm_slab<1, true"]),
    ("`k_gemm_slab<…, GEN>` weight form (token-form weight searches)", ["k_gemm_slab<2, false", "k_gemm_slab<1, false"]),
    ("`k_gemm_stream` (fc2 weight search `MX`, patch embedding)", ["k_gemm_stream"]),
    ("`k_gemm_grpw<GEN>` / `k_gemm_grpk8` / `k_gemm_avq` (attention)", ["k_gemm_grp", "k_gemm_avq", "k_gemm_win"]),
    ("operand packs", ["k_pack"]),
    ("finish + top-k", ["k_finish", "k_topk", "k_fused_finish"]),
    ("`k_score_sorted` + sort + prefix", ["k_score_sorted", "k_sp_", "rocprim", "k_seg_offsets", "k_rs_"]),
    ("radix select + candidate grids", ["k_sel_", "k_candidate_grid"]),
]


def load(path):
    rows = list(csv.DictReader(open(path)))
    return rows, sum(float(x["TotalDurationNs"]) for x in rows)


prod, tp = load(f"profiles/{r}_kernel_stats_deit_small_w4a4.csv")
ref, tr = load(f"profiles/{r}_kernel_stats_deit_small_w4a4_allrounds.csv")
print("| kernel | launches (default / reference schedule) | ms (default / reference) |")
print("|---|---|---|")
seen_p, seen_r = 0.0, 0.0
for name, keys in GROUPS:
    def agg(rows):
        sel = [x for x in rows if any(k in x["Name"] for k in keys)]
        return sum(int(x["Calls"]) for x in sel) / 3, sum(float(x["TotalDurationNs"]) for x in sel) / 3e6
    cp, mp = agg(prod)
    cr, mr = agg(ref)
    seen_p += mp
    seen_r += mr
    print(f"| {name} | {cp:.0f} / {cr:.0f} | {mp:.0f} / {mr:.0f} |")
print(f"| everything else (FP forward passes, copies, fills, small kernels) | | {tp / 3e6 - seen_p:.0f} / {tr / 3e6 - seen_r:.0f} |")
print(f"| **GPU time per calibration** | | **{tp / 3e6:.0f} / {tr / 3e6:.0f}** |")
SCORING = ["k_act_fused_asm", "k_ga_quad", "k_ga_rect", "k_gram_score", "k_gemm_"]
for label, rows in (("default", prod), ("reference", ref)):
    rest = [x for x in rows if not any(k in x["Name"] for k in SCORING)]
    print(f"{label} schedule: {sum(int(x['Calls']) for x in rows) / 3:.0f} launches per calibration; outside the scoring kernels "
          f"{sum(int(x['Calls']) for x in rest) / 3:.0f} launches, {sum(float(x['TotalDurationNs']) for x in rest) / 3e6:.0f} ms")
