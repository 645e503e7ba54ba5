#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4t; mkdir -p $OUT
timeout -k 10 200 python3 tools/probes/probe_short_batch.py 2>&1 | grep -v amdgpu.ids | tee $OUT/short_batch.txt
