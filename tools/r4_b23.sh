#!/bin/bash
# Round 4, batch 23: the force pass's tile at 464 slots (four workgroups of 40 864 B = the 160 KB of a CU) against 448 (HEAD)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4u; mkdir -p $OUT
PROBE_PRE_STEPS=1 timeout -k 10 600 python3 tools/probes/probe_time_kernel.py C5 k_forces 20 3 "@tools/_exp/libsphx_r4p.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/forces_tile_464_c5.txt
timeout -k 10 600 python3 tools/probes/probe_ab_switches.py C5 100 40 1000 300 2 "@tools/_exp/libsphx_r4p.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/forces_tile_464_ab_c5.txt
timeout -k 10 300 python3 tools/probes/probe_ab_switches.py "dp=0.004,DL=20" 300 40 0 0 2 "@tools/_exp/libsphx_r4p.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/forces_tile_464_ab_1p25m.txt
