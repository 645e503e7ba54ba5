#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.txt
timeout -k 10 400 python3 tools/probes/probe_ab_switches.py "dp=0.0045,DL=12" 300 40 2000 600 1 "" tiles_be_from_1000000 2>&1 | grep -v amdgpu.ids | tee $OUT/tiles_from_0p6m_after.txt
