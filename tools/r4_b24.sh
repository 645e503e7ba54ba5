#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4u; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.txt
