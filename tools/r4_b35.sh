#!/bin/bash
# Round 4, batch 35: tiles + slot-coded lists in passes A, B, E below 10^6 particles (SPHX_DEBUG_SWITCHES=tiles_be_from_N), with the
# round-4 walks: C4 (0.5 M) and 0.8 M particles
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 400 python3 tools/probes/probe_ab_switches.py C4 300 40 2000 600 2 "" tiles_be_from_250000 "tiles_be_from_250000,no_fuse_ea" no_fuse_ea 2>&1 | grep -v amdgpu.ids | tee $OUT/tiles_from_c4.txt
timeout -k 10 400 python3 tools/probes/probe_ab_switches.py "dp=0.0045,DL=16" 300 40 2000 600 2 "" tiles_be_from_500000 2>&1 | grep -v amdgpu.ids | tee $OUT/tiles_from_0p8m.txt
