mkdir -p gpurun_out/r6q
python -m pytest tests/test_gpu_e2e.py -x -q -k "full_shape" > gpurun_out/r6q/pytest1.log 2>&1; tail -25 gpurun_out/r6q/pytest1.log
