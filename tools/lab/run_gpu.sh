cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r9
(time timeout 2400 python -m pytest tests -m gpu -x -q) > gpurun_out/r9/pytest.log 2>&1; tail -4 gpurun_out/r9/pytest.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r9/bench.json 2> gpurun_out/r9/bench.err; head -c 250 gpurun_out/r9/bench.json; echo
timeout 300 python bench.py --bits 6 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r9/bench_w6.json 2> gpurun_out/r9/bench_w6.err; head -c 250 gpurun_out/r9/bench_w6.json; echo
