cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r3e
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err; head -c 200 gpurun_out/r3e/bench.json; echo
ADALOG_GEMM_SLAB_FORCE128=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3e/bench_128.json 2> gpurun_out/r3e/bench_128.err; head -c 200 gpurun_out/r3e/bench_128.json; echo
for m in vit_base swin_base; do timeout 300 python bench.py --model $m --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3e/bench_$m.json 2> gpurun_out/r3e/bench_$m.err; head -c 200 gpurun_out/r3e/bench_$m.json; echo; done
