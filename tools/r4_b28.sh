#!/bin/bash
# Round 4, batch 28: C5, static schedule with the wider skins against the device-decided default
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4w; mkdir -p $OUT
timeout -k 10 900 python3 tools/probes/probe_k_skin.py C5 100 40 1000 300 0:0 10:0.42:2 12:0.49:2 8:0.35:2 10:0.42:1 0:0 2>&1 | grep -v amdgpu.ids | tee $OUT/static_vs_dyn_c5.txt
