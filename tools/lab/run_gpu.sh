cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r5
(time timeout 2400 python -m pytest tests -m gpu -x -q) > gpurun_out/r5/pytest.log 2>&1; tail -4 gpurun_out/r5/pytest.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5/bench.json 2> gpurun_out/r5/bench.err; head -c 250 gpurun_out/r5/bench.json; echo
timeout 900 python tools/lab/gram_act_check.py swin > gpurun_out/r5/ga_swin.log 2>&1; tail -9 gpurun_out/r5/ga_swin.log | cut -c1-300
