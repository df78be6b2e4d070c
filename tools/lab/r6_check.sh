# round-6 lab: quant_forward suites + bench line
mkdir -p gpurun_out/r6n
python -m pytest tests/test_gpu_kernels.py -x -q -k "softmax_adalog or gelu_prologue or attn_split or addend or gemm_out_gen or gemm_cand or gemm_score" > gpurun_out/r6n/pytest1.log 2>&1; tail -3 gpurun_out/r6n/pytest1.log
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_golden_forward.py tests/test_gpu_layers.py tests/test_gpu_wrapper.py tests/test_gpu_traces.py -x -q > gpurun_out/r6n/pytest2.log 2>&1; tail -3 gpurun_out/r6n/pytest2.log
python bench.py --steps 1 --warmup 1 --no-rerun-all 2>gpurun_out/r6n/bench.err > gpurun_out/r6n/bench.json; python -c "
import json;d=json.load(open('gpurun_out/r6n/bench.json'));print(round(d['ms_per_step'],1)); q=d['quant_forward']; print({k:q[k] for k in q if k not in ('note','how')})"
