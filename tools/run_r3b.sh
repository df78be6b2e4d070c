cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r3b
(time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "act_gen or sorted") > gpurun_out/r3b/pytest_new.log 2>&1; echo "new rc=$?"; tail -15 gpurun_out/r3b/pytest_new.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err; head -c 400 gpurun_out/r3b/bench.json; echo
ADALOG_GEN_ACT=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3b/bench_nogen.json 2> gpurun_out/r3b/bench_nogen.err; head -c 400 gpurun_out/r3b/bench_nogen.json; echo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3b/prof -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3b/prof.log 2>&1
rm -f gpurun_out/r3b/prof/p_kernel_trace.csv gpurun_out/r3b/prof/*/p_kernel_trace.csv
timeout 300 python tools/lab/find_copies.py > gpurun_out/r3b/copies.txt 2>&1; tail -30 gpurun_out/r3b/copies.txt
(time timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_e2e.py) > gpurun_out/r3b/pytest_gpu.log 2>&1; echo "all rc=$?"; tail -8 gpurun_out/r3b/pytest_gpu.log
