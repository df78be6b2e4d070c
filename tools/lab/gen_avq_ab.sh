# same-box A/B of the log-base search's generated operand (ADALOG_GEN_AVQ) per calibration
for m in deit_small swin_base; do for v in 0 1 0 1; do echo $m GEN_AVQ=$v; ADALOG_GEN_AVQ=$v python bench.py --model $m --steps 2 --warmup 1 --no-cpu-baseline --no-rerun-all 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])
for k in d['config']['scoring_kernels']:
    if 'avq' in k['kernel'] or k['kernel'].startswith('k_gemm_stream<bf16>'): print(k)
"; done; done
