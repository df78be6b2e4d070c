#!/bin/bash
# copies the summaries of tools/final_run.sh (gpurun_out/final, scratch) into profiles/ (tracked), named per round
r=${1:-r06}
f=gpurun_out/final
cp $f/bench.json profiles/${r}_bench.json
for m in deit_tiny vit_base swin_small swin_base w3 w6 swin_base_w3_128img; do [ -f $f/bench_$m.json ] && cp $f/bench_$m.json profiles/${r}_bench_$m.json; done
grep -h '^{' $f/bench_gloo2.json | tail -1 > profiles/${r}_bench_gloo_2ranks_1gpu.json
cp $f/prof/p_kernel_stats.csv profiles/${r}_kernel_stats_deit_small_w4a4.csv
cp $f/prof/p_domain_stats.csv profiles/${r}_domain_stats.csv
[ -f $f/prof_all/p_kernel_stats.csv ] && cp $f/prof_all/p_kernel_stats.csv profiles/${r}_kernel_stats_deit_small_w4a4_allrounds.csv
for m in vit_base swin_base; do [ -f $f/prof_$m/p_kernel_stats.csv ] && cp $f/prof_$m/p_kernel_stats.csv profiles/${r}_kernel_stats_${m}_w4a4.csv; done
[ -f $f/pmc/traffic.json ] && cp $f/pmc/traffic.json profiles/${r}_pmc_bench_traffic.json
[ -f $f/pmc_fused/summary.json ] && cp $f/pmc_fused/summary.json profiles/${r}_pmc_fused_summary.json
[ -f $f/pmc_slab/summary.json ] && cp $f/pmc_slab/summary.json profiles/${r}_pmc_slab_summary.json
[ -f $f/pmc_gram/summary.json ] && cp $f/pmc_gram/summary.json profiles/${r}_pmc_gram_summary.json
[ -f gpurun_out/e2e_outcomes.jsonl ] && cp gpurun_out/e2e_outcomes.jsonl profiles/${r}_e2e_outcomes.jsonl
tail -5 $f/pytest_gpu.log > profiles/${r}_pytest_gpu_tail.log
for n in trace_parity fullshape_parity golden_forward_parity wrapper_flow_parity brecq_traj_parity; do [ -f gpurun_out/$n.jsonl ] && cp gpurun_out/$n.jsonl profiles/${r}_$n.jsonl; done
[ -f gpurun_out/brecq_convergence.json ] && cp gpurun_out/brecq_convergence.json profiles/${r}_brecq_convergence.json
for m in deit_small vit_base; do [ -f $f/kernel_stats_brecq_${m}_block.csv ] && cp $f/kernel_stats_brecq_${m}_block.csv profiles/${r}_kernel_stats_brecq_${m}_block.csv; done
for m in bq_gemm_bench bq_gemm_bench_vit_base; do [ -f $f/$m.json ] && cp $f/$m.json profiles/${r}_$m.json; done
ls -la profiles | grep ${r}_
