# round-6 lab: BRECQ with GELU inside the fc2 input quantiser's kernels and the step counter folded into the Adam kernels
mkdir -p gpurun_out/r6o
python -m pytest tests/test_gpu_kernels.py -x -q -k "adalog or adam or brecq or gelu" > gpurun_out/r6o/pytest1.log 2>&1; tail -3 gpurun_out/r6o/pytest1.log
python -m pytest tests/test_gpu_layers.py -x -q > gpurun_out/r6o/pytest2.log 2>&1; tail -3 gpurun_out/r6o/pytest2.log
for m in deit_small vit_base; do for v in 1 0 1 0; do echo "QF_FUSED=$v $m"; ADALOG_QF_FUSED=$v python tools/bench_brecq.py --model $m --iters 600 2>&1 | grep -v amdgpu.ids | tail -1; done; done
