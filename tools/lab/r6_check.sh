mkdir -p gpurun_out/r6o
python tools/bench_brecq.py --iters 300 --kernel-stats gpurun_out/r6o/ks_fused.csv 2>&1 | grep -v amdgpu | tail -2
ADALOG_BRECQ_SOFTMAX_QUANT=0 python tools/bench_brecq.py --iters 300 --kernel-stats gpurun_out/r6o/ks_unfused.csv 2>&1 | grep -v amdgpu | tail -2
