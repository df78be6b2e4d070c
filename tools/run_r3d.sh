cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r3d
(time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "finish_topk or pack") > gpurun_out/r3d/pytest_new.log 2>&1; echo "new rc=$?"; tail -4 gpurun_out/r3d/pytest_new.log
for i in 1 2; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3d/bench_$i.json 2> gpurun_out/r3d/bench_$i.err; head -c 330 gpurun_out/r3d/bench_$i.json; echo
ADALOG_FINISH_TOPK=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3d/bench_nofuse_$i.json 2> gpurun_out/r3d/bench_nofuse_$i.err; head -c 330 gpurun_out/r3d/bench_nofuse_$i.json; echo
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d/prof -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3d/prof.log 2>&1
rm -f gpurun_out/r3d/prof/p_kernel_trace.csv gpurun_out/r3d/prof/*/p_kernel_trace.csv
(time timeout 2400 python -m pytest tests -m gpu -q) > gpurun_out/r3d/pytest_gpu.log 2>&1; echo "all rc=$?"; tail -8 gpurun_out/r3d/pytest_gpu.log
