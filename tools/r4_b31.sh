#!/bin/bash
# Round 4, batch 31: grid of the self-skipping re-binning launches of a device-decided step: 4096 workgroups (default) / 2048 / 1024
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4x; mkdir -p $OUT
timeout -k 10 900 python3 tools/probes/probe_ab_switches.py C5 100 40 1000 300 2 "" "@tools/_exp/libsphx_dyn2048.so" "@tools/_exp/libsphx_dyn1024.so" 2>&1 | grep -v amdgpu.ids | tee $OUT/dyn_blocks_c5.txt
timeout -k 10 400 python3 tools/probes/probe_ab_switches.py "dp=0.003,DL=22.5" 200 40 1000 300 2 "" "@tools/_exp/libsphx_dyn2048.so" "@tools/_exp/libsphx_dyn1024.so" 2>&1 | grep -v amdgpu.ids | tee $OUT/dyn_blocks_2p5m.txt
