for v in 0 win; do echo GEN_MM=$v; ADALOG_GEN_MM=$v python bench.py --model swin_base --steps 1 --warmup 1 --no-cpu-baseline --no-rerun-all --schedule product 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])
for k in d['config']['scoring_kernels']:
    if 'win' in k['kernel']: print(k)
"; done; for v in win all win all; do echo deit GEN_MM=$v; ADALOG_GEN_MM=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rerun-all 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])
for k in d['config']['scoring_kernels']:
    if 'grpw' in k['kernel']: print(k)
"; done
