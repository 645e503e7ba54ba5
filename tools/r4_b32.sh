#!/bin/bash
# Round 4, batch 32: the cost side of a blocked particle order, measured: decoding a 3 + 14-range layout for every staged slot
# (-DSPHX_EXP_N2_PROLOGUE=1, results unchanged) in all four passes; and both sides together for the force pass (its 320-slot
# tile as if complete + that decode)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4y; mkdir -p $OUT
PROBE_PRE_STEPS=1 timeout -k 10 600 python3 tools/probes/probe_time_kernel.py C5 k_density_walk,k_kgc,k_forces,k_continuity 20 3 "" "@tools/_exp/libsphx_n2cost.so" 2>&1 | grep -v amdgpu.ids | tee $OUT/n2_prologue_cost_c5.txt
PROBE_PRE_STEPS=0 timeout -k 10 600 python3 tools/probes/probe_time_kernel.py C5 k_forces 20 3 "" "forces_tile_320@tools/_exp/libsphx_n2both.so" 2>&1 | grep -v amdgpu.ids | tee $OUT/n2_both_sides_forces_c5.txt
