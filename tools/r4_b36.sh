#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 400 python3 tools/probes/probe_ab_switches.py "dp=0.0045,DL=12" 300 40 2000 600 2 "" tiles_be_from_500000 2>&1 | grep -v amdgpu.ids | tee $OUT/tiles_from_0p6m.txt
timeout -k 10 400 python3 tools/probes/probe_ab_switches.py "dp=0.005,DL=13" 300 40 2000 600 2 "" tiles_be_from_500000 2>&1 | grep -v amdgpu.ids | tee $OUT/tiles_from_0p52m.txt
