cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r3f
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err; head -c 200 gpurun_out/r3f/bench.json; echo
ADALOG_MM_L3_MB=200 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3f/bench_l3.json 2> gpurun_out/r3f/bench_l3.err; head -c 200 gpurun_out/r3f/bench_l3.json; echo
ADALOG_MM_L3_MB=400 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3f/bench_l3b.json 2> gpurun_out/r3f/bench_l3b.err; head -c 200 gpurun_out/r3f/bench_l3b.json; echo
(time timeout 900 python -m pytest tests/test_gpu_fullshape.py tests/test_gpu_e2e.py -m gpu -q -x -k "128img or deit_base") > gpurun_out/r3f/pytest_new.log 2>&1; echo "new rc=$?"; tail -5 gpurun_out/r3f/pytest_new.log
timeout 900 python bench.py --model swin_base --bits 3 --images-per-gpu 128 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r3f/bench_swin_base_w3_128img.json 2> gpurun_out/r3f/bench_swin128.err; head -c 250 gpurun_out/r3f/bench_swin_base_w3_128img.json; echo
ITERS=40 timeout 300 python tools/lab/brecq_ops.py > gpurun_out/r3f/brecq_ops.log 2>&1; head -30 gpurun_out/r3f/brecq_ops.log
