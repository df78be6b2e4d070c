cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r3a
(time timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden_forward.py tests/test_gpu_wrapper.py -m gpu -q -x) > gpurun_out/r3a/pytest_new.log 2>&1; echo "new rc=$?"; tail -15 gpurun_out/r3a/pytest_new.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; head -c 400 gpurun_out/r3a/bench.json; echo
ADALOG_SORTED_SELF=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/bench_old.json 2> gpurun_out/r3a/bench_old.err; head -c 400 gpurun_out/r3a/bench_old.json; echo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3a/prof -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3a/prof.log 2>&1
rm -f gpurun_out/r3a/prof/p_kernel_trace.csv gpurun_out/r3a/prof/*/p_kernel_trace.csv
(time timeout 1500 python -m pytest tests -m gpu -q) > gpurun_out/r3a/pytest_gpu.log 2>&1; echo "all rc=$?"; tail -8 gpurun_out/r3a/pytest_gpu.log
