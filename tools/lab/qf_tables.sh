#!/bin/bash
# Kernel tables of one quant_forward pass (deit_small W4A4, 32 images; rocprofv3 kernel trace of tools/lab/qf_prof.py, 20 passes averaged):
# the fused block route (default) and the module-by-module route (ADALOG_QF_FUSED=0, ADALOG_QF_GEN=0).  -> <out_dir>/qf_table_{fused,modules}.txt
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
out=${1:-gpurun_out/qf_tables}
mkdir -p $out
for v in fused modules; do
  if [ $v = fused ]; then unset ADALOG_QF_FUSED ADALOG_QF_GEN; else export ADALOG_QF_FUSED=0 ADALOG_QF_GEN=0; fi
  rm -rf $out/prof
  QF_REPS=20 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o p -- python3 tools/lab/qf_prof.py > $out/qf_$v.log 2>&1
  f=$(ls $out/prof/*/p_kernel_trace.csv $out/prof/p_kernel_trace.csv 2>/dev/null | head -1)
  python tools/lab/qf_table.py $f 20 grid > $out/qf_table_$v.txt
  rm -rf $out/prof
  echo "== $v"; cat $out/qf_table_$v.txt
done
