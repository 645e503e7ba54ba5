#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_large_configs.py -m gpu -q -x -k "M590k or C4" > $OUT/pytest_m590k.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_m590k.txt
