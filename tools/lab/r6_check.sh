mkdir -p gpurun_out/r6q
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_out_gen" > gpurun_out/r6q/pytest1.log 2>&1; tail -3 gpurun_out/r6q/pytest1.log
