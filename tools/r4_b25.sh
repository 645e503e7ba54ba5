#!/bin/bash
# Round 4, batch 25: the mid-size (index-difference lists) force pass with a tile of 320 (five workgroups per CU), 368 (five, exactly
# the LDS of a CU) and 464 slots (four)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4v; mkdir -p $OUT
PROBE_PRE_STEPS=3 timeout -k 10 300 python3 tools/probes/probe_time_kernel.py C4 k_forces 50 3 "" forces_tile_368 forces_tile_464 2>&1 | grep -v amdgpu.ids | tee $OUT/forces_tile_mid_c4.txt
PROBE_PRE_STEPS=3 timeout -k 10 300 python3 tools/probes/probe_time_kernel.py "dp=0.0042,DL=16" k_forces 50 2 "" forces_tile_368 forces_tile_464 2>&1 | grep -v amdgpu.ids | tee $OUT/forces_tile_mid_0p9m.txt
timeout -k 10 300 python3 tools/probes/probe_ab_switches.py C4 300 40 0 0 2 "" forces_tile_368 forces_tile_464 2>&1 | grep -v amdgpu.ids | tee $OUT/forces_tile_mid_ab_c4.txt
