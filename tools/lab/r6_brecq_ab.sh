# round-6 lab: two-term splits in the BRECQ contractions: convergence at the reference length, BRECQ-only kernel stats
mkdir -p gpurun_out/r6e
for t in 2 3; do
  ADALOG_BRECQ_FWD_TERMS=$t ADALOG_BRECQ_GRAD_TERMS=$t python -m pytest tests/test_gpu_layers.py -x -q -k "converges_at_reference" 2>&1 | tail -1
  cp gpurun_out/brecq_convergence.json gpurun_out/r6e/brecq_convergence_terms$t.json; cat gpurun_out/brecq_convergence.json; echo
done
python tools/bench_brecq.py --model deit_small --iters 1200 --kernel-stats gpurun_out/r6e/kernel_stats_brecq_deit_small_block.csv 2>/dev/null | tail -3
python tools/bench_brecq.py --model vit_base --iters 600 --kernel-stats gpurun_out/r6e/kernel_stats_brecq_vit_base_block.csv 2>/dev/null | tail -3
python -m pytest tests/test_gpu_layers.py tests/test_gpu_kernels.py -x -q -k "brecq or trajectory or f32x3" 2>&1 | tail -2
