# same-box A/B of one environment switch on the default bench line: bash tools/lab/ab_env.sh VAR A B [pairs] [bench args...]
var=$1; a=$2; b=$3; n=${4:-3}; shift 4
for i in $(seq $n); do for v in $a $b; do
  ms=$(env $var=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rerun-all "$@" 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],1))")
  echo "$var=$v $ms"
done; done
