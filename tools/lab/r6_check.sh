# round-6 lab: sort tests, bench and a kernel-stats profile of the current tree
mkdir -p gpurun_out/r6h
python -m pytest tests/test_gpu_kernels.py -x -q -k "sort or sorted or gram_act or self or pack" > gpurun_out/r6h/pytest.log 2>&1; tail -3 gpurun_out/r6h/pytest.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r6h/bench.json 2> gpurun_out/r6h/bench.err
python -c "
import json;d=json.load(open('gpurun_out/r6h/bench.json'));print(d['ms_per_step'], d['config']['other_schedule']['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6h/prof -o ks -- python3 bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-rerun-all > gpurun_out/r6h/bench_prof.json 2> gpurun_out/r6h/prof.err; find gpurun_out/r6h/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r6h/kernel_stats_ref.csv \; ; rm -rf gpurun_out/r6h/prof
