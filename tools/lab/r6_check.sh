# round-6 lab: threshold bisection in k_score_sorted -- tests, then kernel time with / without on one box
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r6p
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_traces.py -x -q -k "sorted or self or trace" > gpurun_out/r6p/pytest1.log 2>&1; tail -3 gpurun_out/r6p/pytest1.log
for v in 1 0; do
  export ADALOG_SS_THR=$v
  rm -rf gpurun_out/r6p/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6p/prof -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rerun-all > gpurun_out/r6p/prof_$v.log 2>&1
  f=$(ls gpurun_out/r6p/prof/*/p_kernel_stats.csv gpurun_out/r6p/prof/p_kernel_stats.csv 2>/dev/null | head -1)
  echo "ADALOG_SS_THR=$v"; grep "k_score_sorted" $f | cut -c1-60,100-200; grep -h '^{' gpurun_out/r6p/prof_$v.log | python -c "import sys,json; print(round(json.loads(sys.stdin.readline())['ms_per_step'],1))"
  rm -f gpurun_out/r6p/prof/*/p_kernel_trace.csv gpurun_out/r6p/prof/p_kernel_trace.csv
done
