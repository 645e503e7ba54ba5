#!/bin/bash
# Round 4, batch 6: full GPU suite + the lean in-place re-binning against round 3
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4g; mkdir -p $OUT
timeout -k 10 420 python3 -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.txt
timeout -k 10 500 python3 tools/probes/probe_ab_switches.py C5 100 40 1000 300 3 "@tools/_exp/libsphx_r3.so" "full_copyback" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c5.txt
