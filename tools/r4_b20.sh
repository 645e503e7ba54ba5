#!/bin/bash
# Round 4, batch 20: the four tile reads of a turn of the near walk requested together: commit f4a881d (near walks) against the working tree
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4s; mkdir -p $OUT
PROBE_PRE_STEPS=1 timeout -k 10 600 python3 tools/probes/probe_time_kernel.py C5 k_density_walk 20 3 "@tools/_exp/libsphx_r4n.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/batched_lds_walk_c5.txt
timeout -k 10 600 python3 tools/probes/probe_ab_switches.py C5 100 40 0 0 3 "@tools/_exp/libsphx_r4n.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/batched_lds_ab_c5.txt
timeout -k 10 300 python3 tools/probes/probe_ab_switches.py "dp=0.004,DL=20" 300 40 0 0 2 "@tools/_exp/libsphx_r4n.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/batched_lds_ab_1p25m.txt
timeout -k 10 400 python3 -m pytest tests/test_gpu_large_configs.py tests/test_gpu_grid_skin.py tests/test_slab.py -m gpu -q -x > $OUT/pytest_sub.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_sub.txt
