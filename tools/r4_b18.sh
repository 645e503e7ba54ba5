#!/bin/bash
# Round 4, batch 18: walks that read the tile without looking at the entry where the tile holds the workgroup's whole neighbourhood
# (span <= tile size) -- per pass and per step against the build before (tools/_exp/libsphx_r4k.so = commit cbfcec3), then the suite
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4r; mkdir -p $OUT
PROBE_PRE_STEPS=1 timeout -k 10 600 python3 tools/probes/probe_time_kernel.py C5 k_density_walk,k_kgc,k_forces,k_continuity 20 2 "@tools/_exp/libsphx_r4k.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/near_walk_per_pass_c5.txt
timeout -k 10 900 python3 tools/probes/probe_ab_switches.py C5 100 40 1000 300 2 "@tools/_exp/libsphx_r4k.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/near_walk_ab_c5.txt
timeout -k 10 300 python3 tools/probes/probe_ab_switches.py "dp=0.004,DL=20" 300 40 1000 300 2 "@tools/_exp/libsphx_r4k.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/near_walk_ab_1p25m.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.txt
