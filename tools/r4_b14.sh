#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4o; mkdir -p $OUT
timeout -k 10 400 python3 tools/probes/probe_k_skin.py C2 4000 400 20000 12000 16:0 20:1.05 24:1.05 24:1.2 28:1.2 20:0.9 16:0.8 24:1.05 16:0 2>&1 | grep -v amdgpu.ids | tee $OUT/k_skin_c2_b.txt
timeout -k 10 300 python3 tools/probes/probe_k_skin.py C1 4000 400 20000 12000 16:0 24:1.05 24:1.2 16:0.8 16:0 2>&1 | grep -v amdgpu.ids | tee $OUT/k_skin_c1.txt
timeout -k 10 300 python3 tools/probes/probe_k_skin.py "dp=0.02,DL=4" 4000 400 20000 12000 16:0 24:1.05 24:1.2 16:0.8 2>&1 | grep -v amdgpu.ids | tee $OUT/k_skin_dp002.txt
timeout -k 10 300 python3 tools/probes/probe_k_skin.py "dp=0.015,DL=3" 4000 400 20000 8000 8:0 12:0.49 12:0.6 8:0.42 2>&1 | grep -v amdgpu.ids | tee $OUT/k_skin_dp0015.txt
