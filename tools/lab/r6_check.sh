# round-6 lab: softmax pack kernel after the rows-per-wave change
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r6m gpurun_out/r6n
python -m pytest tests/test_gpu_kernels.py -x -q -k "softmax_adalog" > gpurun_out/r6n/pytest1.log 2>&1; tail -3 gpurun_out/r6n/pytest1.log
rm -rf gpurun_out/r6m/qf
QF_REPS=20 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6m/qf -o p -- python3 tools/lab/qf_prof.py > gpurun_out/r6m/qf.log 2>&1
f=$(ls gpurun_out/r6m/qf/*/p_kernel_trace.csv gpurun_out/r6m/qf/p_kernel_trace.csv 2>/dev/null | head -1)
python tools/lab/qf_table.py $f 20 | tee gpurun_out/r6m/qf_table_fused.txt
rm -f $f
