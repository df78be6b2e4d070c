cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r3c
(time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "finish_topk or pack or act_gen") > gpurun_out/r3c/pytest_new.log 2>&1; echo "new rc=$?"; tail -6 gpurun_out/r3c/pytest_new.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err; head -c 330 gpurun_out/r3c/bench.json; echo
ADALOG_SLAB_DYN=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3c/bench_dyn.json 2> gpurun_out/r3c/bench_dyn.err; head -c 330 gpurun_out/r3c/bench_dyn.json; echo
ADALOG_FINISH_TOPK=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3c/bench_nofuse.json 2> gpurun_out/r3c/bench_nofuse.err; head -c 330 gpurun_out/r3c/bench_nofuse.json; echo
ADALOG_PACK_TAB=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3c/bench_notab.json 2> gpurun_out/r3c/bench_notab.err; head -c 330 gpurun_out/r3c/bench_notab.json; echo
(ADALOG_SLAB_DYN=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_traces.py tests/test_gpu_fullshape.py -m gpu -q -x -k "gemm or slab or act_gen or linear or finish") > gpurun_out/r3c/pytest_dyn.log 2>&1; echo "dyn rc=$?"; tail -6 gpurun_out/r3c/pytest_dyn.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3c/prof -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3c/prof.log 2>&1
rm -f gpurun_out/r3c/prof/p_kernel_trace.csv gpurun_out/r3c/prof/*/p_kernel_trace.csv
(time timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_e2e.py) > gpurun_out/r3c/pytest_gpu.log 2>&1; echo "all rc=$?"; tail -8 gpurun_out/r3c/pytest_gpu.log
