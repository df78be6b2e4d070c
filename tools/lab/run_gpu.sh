cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r8
for v in l0 new l0 new; do
  echo "variant $v"
  if [ $v = new ]; then L=$PWD/adalog_amd/csrc/libadalog_hip.so; else L=$PWD/tools/lab/variants/libadalog_$v.so; fi
  ADALOG_LIB=$L timeout 300 python tools/lab/gram_act_check.py x 2>&1 | grep shape | cut -c1-260
done
ADALOG_LIB=$PWD/tools/lab/variants/libadalog_l0.so timeout 300 python tools/lab/gram_act_check.py split 2>&1 | grep shape | cut -c1-260 | head -5
timeout 300 python tools/lab/gram_act_check.py split 2>&1 | grep shape | cut -c1-260 | head -5
timeout 600 python -m pytest tests -m gpu -x -q -k "gram_act" 2>&1 | tail -2
