cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r1
timeout 600 python tools/lab/gram_act_check.py split > gpurun_out/r1/ga_split.log 2>&1; tail -12 gpurun_out/r1/ga_split.log
timeout 900 python -m pytest tests -m gpu -x -q -k "gram_act or brecq or train_mm or adaround" > gpurun_out/r1/pytest.log 2>&1; tail -5 gpurun_out/r1/pytest.log
for o in 0 1; do ADALOG_BQ_OVERLAP=$o timeout 300 python tools/bench_brecq.py --iters 300 > gpurun_out/r1/brecq_ov$o.log 2>&1; tail -3 gpurun_out/r1/brecq_ov$o.log; done
for o in 0 1; do ADALOG_BQ_OVERLAP=$o timeout 300 python tools/bench_brecq.py --model vit_base --iters 200 > gpurun_out/r1/brecq_vb_ov$o.log 2>&1; tail -2 gpurun_out/r1/brecq_vb_ov$o.log; done
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r1/bench.json 2> gpurun_out/r1/bench.err; head -c 250 gpurun_out/r1/bench.json; echo
ADALOG_LANES=2 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r1/bench_lanes2.json 2> gpurun_out/r1/bench_lanes2.err; head -c 250 gpurun_out/r1/bench_lanes2.json; echo
for m in vit_base swin_base; do timeout 300 python bench.py --model $m --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r1/bench_$m.json 2> gpurun_out/r1/bench_$m.err; head -c 250 gpurun_out/r1/bench_$m.json; echo; done
