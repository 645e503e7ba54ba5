#!/bin/bash
# Round 4, batch 3: lane sorting + the repaired force-pass epilogue against round 3, on one box; the 4-slab ring case five times
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4d; mkdir -p $OUT
for k in 1 2 3 4 5; do timeout -k 10 120 python3 -m pytest tests/test_slab.py -m gpu -x -q -k "native_ring and 0.01-6.0" > $OUT/slab_rep$k.txt 2>&1; echo "slab ring rep $k rc=$?"; tail -1 $OUT/slab_rep$k.txt; done
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "large_configs or grid_skin or headline or resident or edge" > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.txt
timeout -k 10 500 python3 tools/probes/probe_ab_switches.py C5 100 40 1000 300 2 "@tools/_exp/libsphx_r3.so" "no_lane_sort" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c5.txt
timeout -k 10 300 python3 tools/probes/probe_ab_switches.py C4 300 40 2000 1000 2 "@tools/_exp/libsphx_r3.so" "no_lane_sort" "" "@tools/_exp/libsphx_w4.so" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c4.txt
timeout -k 10 100 python3 tools/probes/probe_ab_switches.py C3 2000 100 0 0 2 "@tools/_exp/libsphx_r3.so" "no_lane_sort" "" "@tools/_exp/libsphx_w4.so" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c3.txt
timeout -k 10 100 python3 tools/probes/probe_ab_switches.py C2 4000 400 0 0 3 "@tools/_exp/libsphx_r3.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c2.txt
