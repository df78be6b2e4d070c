#!/bin/bash
# End-of-round validation on the GPU box (run from the repo root, e.g. gpurun -- 'bash tools/final_run.sh'):
# the -m gpu test suite, the default bench line, a rocprofv3 kernel-stats pass, the PMC traffic passes, the 2-rank gloo
# launch path, and one calibration each of the other model families / bit widths.  Everything lands under
# gpurun_out/final/; the summaries that are kept go to profiles/r06_* (tools/collect_profiles.sh r06).
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/final
rm -f gpurun_out/trace_parity.jsonl gpurun_out/fullshape_parity.jsonl gpurun_out/golden_forward_parity.jsonl gpurun_out/wrapper_flow_parity.jsonl gpurun_out/brecq_traj_parity.jsonl
(time timeout 2400 python -m pytest tests -m gpu -q) > gpurun_out/final/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/final/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; head -c 300 gpurun_out/final/bench.json; echo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --schedule product --no-rerun-all > gpurun_out/final/prof.log 2>&1
rm -f gpurun_out/final/prof/p_kernel_trace.csv gpurun_out/final/prof/*/p_kernel_trace.csv
# the same three calibrations on the reference's schedule (every search of every round: what `value` is measured on)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_all -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --schedule reference --no-rerun-all > gpurun_out/final/prof_all.log 2>&1
rm -f gpurun_out/final/prof_all/p_kernel_trace.csv gpurun_out/final/prof_all/*/p_kernel_trace.csv
timeout 600 bash tools/pmc_bench.sh gpurun_out/final/pmc > gpurun_out/final/pmc.log 2>&1; tail -5 gpurun_out/final/pmc.log
timeout 300 bash tools/pmc_fused.sh gpurun_out/final/pmc_fused > gpurun_out/final/pmc_fused.log 2>&1
PROF_DT=fp8 timeout 600 bash tools/pmc_gemm.sh gpurun_out/final/pmc_slab "qkv fc1" > gpurun_out/final/pmc_slab.log 2>&1
timeout 600 bash tools/pmc_gram.sh gpurun_out/final/pmc_gram > gpurun_out/final/pmc_gram.log 2>&1
ADALOG_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/bench_gloo2.json 2> gpurun_out/final/bench_gloo2.err; head -c 200 gpurun_out/final/bench_gloo2.json; echo
for m in deit_tiny vit_base swin_small swin_base; do timeout 300 python bench.py --model $m --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_$m.json 2> gpurun_out/final/bench_$m.err; head -c 200 gpurun_out/final/bench_$m.json; echo; done
for b in 3 6; do timeout 300 python bench.py --bits $b --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_w${b}.json 2> gpurun_out/final/bench_w${b}.err; head -c 200 gpurun_out/final/bench_w${b}.json; echo; done
timeout 900 python bench.py --model swin_base --bits 3 --images-per-gpu 128 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/bench_swin_base_w3_128img.json 2> gpurun_out/final/bench_swin128.err; head -c 200 gpurun_out/final/bench_swin_base_w3_128img.json; echo
# per-model kernel statistics of the two larger BASELINE models (one warm calibration each)
for m in vit_base swin_base; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_$m -o p -- python3 bench.py --model $m --steps 1 --warmup 1 --no-cpu-baseline --schedule product --no-rerun-all > gpurun_out/final/prof_$m.log 2>&1
  rm -f gpurun_out/final/prof_$m/p_kernel_trace.csv gpurun_out/final/prof_$m/*/p_kernel_trace.csv
done
# BRECQ: kernel statistics of the reconstruct_single_block call alone (300 iterations, graph replay; torch.profiler around the call:
# rocprofv3 around the whole script would mix in the FP model's forward and the block's calibration), then the products one by one
timeout 400 python tools/bench_brecq.py --iters 300 --kernel-stats gpurun_out/final/kernel_stats_brecq_deit_small_block.csv > gpurun_out/final/prof_brecq.log 2>&1; tail -2 gpurun_out/final/prof_brecq.log
timeout 400 python tools/bench_brecq.py --model vit_base --iters 300 --kernel-stats gpurun_out/final/kernel_stats_brecq_vit_base_block.csv > gpurun_out/final/prof_brecq_vit_base.log 2>&1; tail -2 gpurun_out/final/prof_brecq_vit_base.log
timeout 300 python tools/lab/bq_gemm_bench.py --terms 2 --int-act --out gpurun_out/final/bq_gemm_bench.json > gpurun_out/final/bq_gemm_bench.log 2>&1; tail -2 gpurun_out/final/bq_gemm_bench.log
timeout 300 python tools/lab/bq_gemm_bench.py --model vit_base --terms 2 --int-act --out gpurun_out/final/bq_gemm_bench_vit_base.json > gpurun_out/final/bq_gemm_bench_vit_base.log 2>&1; tail -1 gpurun_out/final/bq_gemm_bench_vit_base.log
