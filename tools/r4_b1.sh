#!/bin/bash
# Round 4, batch 1: the GPU suite on the new re-binning chain / drift bound, then C5 / C4 A/B of the switches on one box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4b; mkdir -p $OUT
timeout -k 10 420 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.txt
timeout -k 10 500 python3 tools/probes/probe_ab_switches.py C5 100 40 1000 300 2 "no_drift_top2,no_sched_redirect" "no_sched_redirect" "no_drift_top2" "" "walk_superset" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c5.txt
timeout -k 10 200 python3 tools/probes/probe_ab_switches.py C4 300 40 2000 1000 2 "no_drift_top2" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c4.txt
