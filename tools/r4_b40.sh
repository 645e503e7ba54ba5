#!/bin/bash
# Round 4, batch 40: lanes per particle around the policy's boundaries (8 up to 33 k, 4 up to 220 k, 2 above)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4z; mkdir -p $OUT
(timeout -k 10 200 python3 tools/probes/probe_lpp.py C3 2000 100 0 2 8
 timeout -k 10 200 python3 tools/probes/probe_lpp.py "dp=0.008,DL=8" 1500 100 0 2
 timeout -k 10 200 python3 tools/probes/probe_lpp.py "dp=0.0065,DL=8.45" 1000 100 0 2
 timeout -k 10 200 python3 tools/probes/probe_lpp.py "dp=0.006,DL=9" 1000 100 0 4
 timeout -k 10 200 python3 tools/probes/probe_lpp.py "dp=0.014,DL=4.2" 3000 100 0 4 16) 2>&1 | grep -v amdgpu.ids | tee $OUT/lpp_policy.txt
