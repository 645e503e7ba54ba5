#!/bin/bash
# Round 4, batch 43: the fused E|A launch beyond 4096 workgroups per pass (tail_limit_8192) against tiles + coded lists
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4z; mkdir -p $OUT
(timeout -k 10 400 python3 tools/probes/probe_ab_switches.py "dp=0.0045,DL=12" 300 40 2000 600 2 "" tail_limit_8192
 timeout -k 10 400 python3 tools/probes/probe_ab_switches.py "dp=0.0045,DL=16" 300 40 2000 600 2 "" tail_limit_8192) 2>&1 | grep -v amdgpu.ids | tee $OUT/fused_limit.txt
