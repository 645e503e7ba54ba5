#!/bin/bash
# Round 4, batch 17: what the far path of the slot-coded fetch costs every pass -- default library against the timing-only build
# whose fetches are LDS reads only (tools/probes/build_variant_lib.sh pretend -DSPHX_EXP_PRETEND_COMPLETE_TILE)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4q; mkdir -p $OUT
PROBE_PRE_STEPS=1 timeout -k 10 900 python3 tools/probes/probe_time_kernel.py C5 k_density_walk 20 3 "" "@tools/_exp/libsphx_pretendA.so" 2>&1 | grep -v amdgpu.ids | tee $OUT/pretend_lds_only_fetch_walk_c5.txt
