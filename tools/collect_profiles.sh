#!/bin/bash
# copies the summaries of tools/final_run.sh (gpurun_out/final, scratch) into profiles/ (tracked), named per round
r=${1:-r02}
f=gpurun_out/final
cp $f/bench.json profiles/${r}_bench.json
for m in deit_tiny vit_base swin_small swin_base w3 w6; do cp $f/bench_$m.json profiles/${r}_bench_$m.json; done
grep -h '^{' $f/bench_gloo2.json | tail -1 > profiles/${r}_bench_gloo_2ranks_1gpu.json
cp $f/prof/p_kernel_stats.csv profiles/${r}_kernel_stats_deit_small_w4a4.csv
cp $f/prof/p_domain_stats.csv profiles/${r}_domain_stats.csv
for m in vit_base swin_base; do [ -f $f/prof_$m/p_kernel_stats.csv ] && cp $f/prof_$m/p_kernel_stats.csv profiles/${r}_kernel_stats_${m}_w4a4.csv; done
cp $f/pmc/traffic.json profiles/${r}_pmc_bench_traffic.json
cp $f/pmc_fused/summary.json profiles/${r}_pmc_fused_summary.json
cp $f/bench_fused.txt profiles/${r}_bench_fused.txt
tail -5 $f/pytest_gpu.log > profiles/${r}_pytest_gpu_tail.log
cp gpurun_out/trace_parity.jsonl profiles/${r}_trace_parity.jsonl
cp gpurun_out/fullshape_parity.jsonl profiles/${r}_fullshape_parity.jsonl
[ -f gpurun_out/brecq_convergence.json ] && cp gpurun_out/brecq_convergence.json profiles/${r}_brecq_convergence.json
ls -la profiles | grep ${r}_
