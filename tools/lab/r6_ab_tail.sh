# round-6 lab: GPU tests touched by the FPCS-tail work, then a same-box A/B of ADALOG_FUSED_TAIL and a kernel-stats profile
mkdir -p gpurun_out/r6d
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_traces.py tests/test_gpu_layers.py tests/test_gpu_distributed.py -x -q > gpurun_out/r6d/pytest.log 2>&1; tail -3 gpurun_out/r6d/pytest.log
for i in 1 2; do for v in 1 0; do
  ADALOG_FUSED_TAIL=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6d/b_${v}_$i.json 2>/dev/null
  python -c "
import json,sys;d=json.load(open('gpurun_out/r6d/b_${v}_$i.json'));print('FUSED_TAIL=$v', round(d['ms_per_step'],1), round(d['config']['other_schedule']['ms_per_step'],1))"
done; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6d/prof -o ks -- python3 bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-rerun-all > gpurun_out/r6d/bench_prof.json 2> gpurun_out/r6d/prof.err; find gpurun_out/r6d/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r6d/kernel_stats_ref.csv \; ; rm -rf gpurun_out/r6d/prof
