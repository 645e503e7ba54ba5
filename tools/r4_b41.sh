#!/bin/bash
# Round 4, batch 41: the compiler's scheduling strategy (-mllvm -amdgpu-sched-strategy=max-ilp | max-memory-clause) against the default
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4z; mkdir -p $OUT
(PROBE_PRE_STEPS=1 timeout -k 10 600 python3 tools/probes/probe_time_kernel.py C5 k_density_walk,k_kgc,k_forces,k_continuity 20 2 "" "@tools/_exp/libsphx_maxilp.so" "@tools/_exp/libsphx_maxmem.so"
 timeout -k 10 300 python3 tools/probes/probe_ab_switches.py C2 4000 400 0 0 2 "" "@tools/_exp/libsphx_maxilp.so" "@tools/_exp/libsphx_maxmem.so"
 timeout -k 10 300 python3 tools/probes/probe_ab_switches.py C4 300 40 0 0 2 "" "@tools/_exp/libsphx_maxilp.so" "@tools/_exp/libsphx_maxmem.so") 2>&1 | grep -v amdgpu.ids | tee $OUT/sched_strategy.txt
