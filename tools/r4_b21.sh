#!/bin/bash
# Round 4, batch 21: the two-stream slab step only from 150 k particles per slab (SPHX_SLAB_OVERLAP=always|never for the tests)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4t; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_slab.py -m gpu -q -x > $OUT/pytest_slab.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_slab.txt
for a in "C5 8 40" "C5 2 40" "C4 2 100" "C2 2 400" "C5 8 40 one-stream"; do
  for ov in never always; do echo -n "[SPHX_SLAB_OVERLAP=$ov] "; SPHX_SLAB_OVERLAP=$ov timeout -k 10 200 python3 tools/probes/probe_slab_ring.py $a 2>&1 | grep -v amdgpu.ids; done
done | tee $OUT/slab_ring.txt
timeout -k 10 200 python3 tools/probes/probe_slab_ring.py C2 2 400 2>&1 | grep -v amdgpu.ids | tee -a $OUT/slab_ring.txt
