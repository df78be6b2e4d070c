# round-6 lab: GENA tests
mkdir -p gpurun_out/r6j
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_out_gen" > gpurun_out/r6j/pytest.log 2>&1; tail -5 gpurun_out/r6j/pytest.log
python -m pytest tests/test_gpu_golden_forward.py -x -q > gpurun_out/r6j/pytest2.log 2>&1; tail -3 gpurun_out/r6j/pytest2.log
