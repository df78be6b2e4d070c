# same-box A/B of one environment switch on the default bench line: bash tools/lab/ab_env.sh VAR A B [pairs] [kernel-substring] [bench args...]
var=$1; a=$2; b=$3; n=${4:-3}; pat=${5:-}; shift 5 2>/dev/null || shift $#
for i in $(seq $n); do for v in $a $b; do
  env $var=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rerun-all "$@" 2>/dev/null | PAT="$pat" V="$var=$v" python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); pat=os.environ['PAT']
ks=' '.join('%s=%.2f' % (k['kernel'], k['ms_per_step']) for k in d['config']['scoring_kernels'] if pat and pat in k['kernel'])
print(os.environ['V'], round(d['ms_per_step'],1), ks)"
done; done
