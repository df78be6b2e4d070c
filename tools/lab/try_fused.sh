#!/bin/bash
# usage (from adalog_amd/csrc): ../../tools/lab/try_fused.sh <extra -D flags>   -- register / scratch report of k_act_fused<12>
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused -Rpass-analysis=kernel-resource-usage "$@" -c gemm_fused.hip -o /tmp/gf.o 2>&1 | grep -A9 "k_act_fusedILi12ELi4" | grep -E "VGPRs|AGPRs|Scratch|Spill|Occupancy"
