# round-6 lab: the BRECQ products as an iteration issues them (two-term operands, integer activations), per tile shape
mkdir -p gpurun_out/r6k
for m in deit_small vit_base; do
  for sh in default 0 1 2; do
    if [ $sh = default ]; then unset ADALOG_BQ_SHAPE; else export ADALOG_BQ_SHAPE=$sh; fi
    python tools/lab/bq_gemm_bench.py --model $m --terms 2 --int-act --out gpurun_out/r6k/bq_${m}_shape_$sh.json > gpurun_out/r6k/bq_${m}_shape_$sh.log 2>&1
    echo "== $m shape $sh"; grep -v amdgpu.ids gpurun_out/r6k/bq_${m}_shape_$sh.log
  done
done
unset ADALOG_BQ_SHAPE
for sh in 0 1; do
  for m in deit_small vit_base; do
    ADALOG_BQ_SHAPE=$sh python tools/bench_brecq.py --model $m --iters 300 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
