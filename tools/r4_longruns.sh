#!/bin/bash
# Round-4 physical runs with the final build: C3 and C4 from rest to t = 20 s, C5 from rest to t = 1 s
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4long; mkdir -p $OUT
timeout -k 10 120 python3 tests/longrun_configs.py C3 20 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/C3.txt
timeout -k 10 300 python3 tests/longrun_configs.py C4 20 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/C4.txt
timeout -k 10 600 python3 tools/probes/probe_soak.py 1.0 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/soak_C5_1s.txt
