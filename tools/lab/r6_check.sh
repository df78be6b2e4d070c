# final tree: whole GPU suite + the default bench line
mkdir -p gpurun_out/final
(time timeout 2400 python -m pytest tests -m gpu -q) > gpurun_out/final/pytest_gpu.log 2>&1; tail -4 gpurun_out/final/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; head -c 300 gpurun_out/final/bench.json; echo
