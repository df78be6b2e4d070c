#!/bin/bash
python tools/gen_fused_asm.py > /dev/null
make -C adalog_amd/csrc 2>&1 | grep -E "error:" && exit 1
echo "== packed"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k fused 2>&1 | tail -2
timeout 60 python tools/bench_fused.py 2>&1 | grep -E "fused=True"
