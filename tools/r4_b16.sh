#!/bin/bash
# Round 4, batch 16: the prize of a blocked particle order for the force pass, measured with a timing-only build whose 320-slot
# tile behaves as if it were complete (tools/probes/build_variant_lib.sh pretend320 -DSPHX_EXP_PRETEND_COMPLETE_TILE)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4q; mkdir -p $OUT
PROBE_PRE_STEPS=0 timeout -k 10 900 python3 tools/probes/probe_time_kernel.py C5 k_forces 20 3 "" forces_tile_320 "forces_tile_320@tools/_exp/libsphx_pretend320.so" "@tools/_exp/libsphx_pretend320.so" 2>&1 | grep -v amdgpu.ids | tee $OUT/pretend_complete_tile_c5.txt
