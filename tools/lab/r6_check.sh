mkdir -p gpurun_out/r6r
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_f32x3" > gpurun_out/r6r/pytest1.log 2>&1; tail -3 gpurun_out/r6r/pytest1.log
python -m pytest tests/test_gpu_layers.py -x -q > gpurun_out/r6r/pytest2.log 2>&1; tail -3 gpurun_out/r6r/pytest2.log
for m in deit_small vit_base; do for v in 1 0 1 0; do echo "ADDEND=$v $m"; ADALOG_BRECQ_ADDEND=$v python tools/bench_brecq.py --model $m --iters 600 2>&1 | grep -v amdgpu.ids | tail -1; done; done
