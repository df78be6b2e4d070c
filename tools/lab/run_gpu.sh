cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r11
timeout 600 python -m pytest tests -m gpu -x -q -k "gram or trace" 2>&1 | tail -2
for v in old new old new old new; do
  cp tools/lab/variants/libadalog_$v.so adalog_amd/csrc/libadalog_hip.so
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r11/bench_${v}.json 2> gpurun_out/r11/bench_${v}.err
  python - gpurun_out/r11/bench_${v}.json $v <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(b['ms_per_step'],1), round(b['config']['other_schedule']['ms_per_step'],1))
PY
done
cp tools/lab/variants/libadalog_new.so adalog_amd/csrc/libadalog_hip.so
timeout 300 python tools/lab/gram_act_check.py x 2>&1 | grep shape | cut -c1-260
