cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests -m gpu -x -q -k "gram_act or sorted or self or trace" > gpurun_out/r2/pytest.log 2>&1; tail -3 gpurun_out/r2/pytest.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/prof -o p -- python3 tools/lab/gram_act_check.py split > gpurun_out/r2/ga_split.log 2>&1; tail -9 gpurun_out/r2/ga_split.log | cut -c1-300
rm -f gpurun_out/r2/prof/p_kernel_trace.csv gpurun_out/r2/prof/*/p_kernel_trace.csv
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err; head -c 250 gpurun_out/r2/bench.json; echo
