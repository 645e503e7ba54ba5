#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4w; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.txt
timeout -k 10 300 python3 tools/probes/probe_k_skin.py "dp=0.0035,DL=20" 300 40 1000 300 0:0 0:0:1 2>&1 | grep -v amdgpu.ids | tee $OUT/static_vs_dyn_1p6m.txt
