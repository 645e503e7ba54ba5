#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4l; mkdir -p $OUT
timeout -k 10 900 python3 tools/probes/probe_k_skin.py C5 100 40 1000 300 8:0.28 10:0.28 12:0.28 16:0.28 10:0.25 12:0.31 8:0.28 5:0 2>&1 | grep -v amdgpu.ids | tee $OUT/k_skin_c5_b.txt
timeout -k 10 400 python3 tools/probes/probe_k_skin.py C4 300 40 2000 1000 5:0 10:0.42 12:0.49 8:0.28:1 12:0.28:1 10:0.35:1 16:0.28:1 5:0:1 5:0 2>&1 | grep -v amdgpu.ids | tee $OUT/k_skin_c4_b.txt
timeout -k 10 300 python3 tools/probes/probe_k_skin.py "dp=0.004,DL=20" 200 40 1500 600 5:0 8:0.28 12:0.28 2>&1 | grep -v amdgpu.ids | tee $OUT/k_skin_1p25m.txt
