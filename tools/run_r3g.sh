cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r3g
(time timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "act_gen or finish_topk") > gpurun_out/r3g/pytest_new.log 2>&1; echo "new rc=$?"; tail -4 gpurun_out/r3g/pytest_new.log
for i in 1 2; do
ADALOG_TORCH_OPS=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3g/bench_new_$i.json 2> gpurun_out/r3g/bench_new_$i.err; head -c 200 gpurun_out/r3g/bench_new_$i.json; echo
ADALOG_TORCH_OPS=0 ADALOG_LIB=$PWD/tools/lab/libadalog_hip_gen_after_barrier.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3g/bench_old_$i.json 2> gpurun_out/r3g/bench_old_$i.err; head -c 200 gpurun_out/r3g/bench_old_$i.json; echo
done
