#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4n; mkdir -p $OUT
timeout -k 10 500 python3 tools/probes/probe_k_skin.py C5 100 40 1000 300 12:0.28 16:0.28 24:0.28 64:0.28 16:0.31 24:0.34 12:0.28 2>&1 | grep -v amdgpu.ids | tee $OUT/k_skin_c5_c.txt
timeout -k 10 300 python3 tools/probes/probe_k_skin.py C3 2000 100 20000 4000 8:0 8:0.35 8:0.42 12:0.49 12:0.42 16:0.6 16:0.49 8:0 2>&1 | grep -v amdgpu.ids | tee $OUT/k_skin_c3.txt
timeout -k 10 300 python3 tools/probes/probe_k_skin.py C2 4000 400 20000 8000 16:0 16:0.7 16:0.5 24:1.05 24:0.7 32:1.05 32:0.7 16:0 2>&1 | grep -v amdgpu.ids | tee $OUT/k_skin_c2.txt
timeout -k 10 300 python3 tools/probes/probe_k_skin.py "dp=0.01,DL=18" 1000 100 6000 3000 8:0 8:0.35 12:0.49 12:0.42 2>&1 | grep -v amdgpu.ids | tee $OUT/k_skin_194k.txt
