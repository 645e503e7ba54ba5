#!/bin/bash
# Round 4, batch 27: where does the device-decided re-binning start to pay?  1.25 M and 2.5 M particles, static schedule (host
# re-bins; K = 10 / 0.42 h and K = 12 / 0.49 h) against the device-decided default (K = 24 / 0.28 h)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4w; mkdir -p $OUT
timeout -k 10 500 python3 tools/probes/probe_k_skin.py "dp=0.004,DL=20" 300 40 1000 400 0:0 10:0.42:2 12:0.49:2 8:0.35:2 0:0 2>&1 | grep -v amdgpu.ids | tee $OUT/static_vs_dyn_1p25m.txt
timeout -k 10 500 python3 tools/probes/probe_k_skin.py "dp=0.003,DL=22.5" 200 40 1000 300 0:0 10:0.42:2 12:0.49:2 0:0 2>&1 | grep -v amdgpu.ids | tee $OUT/static_vs_dyn_2p5m.txt
