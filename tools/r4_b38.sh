#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4z; mkdir -p $OUT
for sw in "" tiles_be_from_200000; do echo -n "[$sw] "; SPHX_DEBUG_SWITCHES=$sw timeout -k 10 200 python3 tools/probes/probe_slab_ring.py C4 2 100 2>&1 | grep -v amdgpu.ids; done | tee $OUT/slab_tiles_from.txt
for sw in "" tiles_be_from_300000; do echo -n "[$sw] "; SPHX_DEBUG_SWITCHES=$sw timeout -k 10 200 python3 tools/probes/probe_slab_ring.py C5 16 40 2>&1 | grep -v amdgpu.ids; done | tee -a $OUT/slab_tiles_from.txt
