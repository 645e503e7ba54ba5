#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4j; mkdir -p $OUT
timeout -k 10 420 python3 -m pytest tests/test_slab.py -m gpu -q > $OUT/pytest_slab.txt 2>&1; echo "pytest slab rc=$?"; tail -3 $OUT/pytest_slab.txt
for a in "C5 8 40" "C5 2 40" "C4 2 100" "C2 2 400" "C5 8 40 one-stream"; do
  for sw in no_slab_overlap ""; do echo -n "[$sw] "; SPHX_DEBUG_SWITCHES=$sw timeout -k 10 200 python3 tools/probes/probe_slab_ring.py $a 2>&1 | grep -v amdgpu.ids; done
done | tee $OUT/slab_ring.txt
