mkdir -p gpurun_out/r6l
timeout 500 tools/lab/bq_lab -1 20 22 sweep | tee gpurun_out/r6l/bq_sweep.txt
