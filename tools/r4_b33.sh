#!/bin/bash
# Round 4, batch 33: a thread's two staged slots requested together (one memory round trip less in every pass's prologue) against
# the staging loop (HEAD = tools/_exp/libsphx_r4q.so)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4z; mkdir -p $OUT
PROBE_PRE_STEPS=1 timeout -k 10 600 python3 tools/probes/probe_time_kernel.py C5 k_density_walk,k_kgc,k_forces,k_continuity 20 3 "@tools/_exp/libsphx_r4q.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/two_slot_staging_per_pass_c5.txt
timeout -k 10 900 python3 tools/probes/probe_ab_switches.py C5 100 40 1000 300 2 "@tools/_exp/libsphx_r4q.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/two_slot_staging_ab_c5.txt
timeout -k 10 300 python3 tools/probes/probe_ab_switches.py C4 300 40 0 0 2 "@tools/_exp/libsphx_r4q.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/two_slot_staging_ab_c4.txt
timeout -k 10 300 python3 tools/probes/probe_ab_switches.py C3 2000 100 0 0 2 "@tools/_exp/libsphx_r4q.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/two_slot_staging_ab_c3.txt
