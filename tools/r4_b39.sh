#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_slab.py -m gpu -q -x > $OUT/pytest_slab.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_slab.txt
