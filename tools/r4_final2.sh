#!/bin/bash
# Round-4 evidence, part 2: PMC passes (C2, C4, C5), slab-ring probes, the short-batch probe
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4final; mkdir -p $OUT
bash tools/probes/profile_pmc.sh C2 0 200 r4f > $OUT/pmc_c2.log 2>&1; echo "pmc C2 done"
bash tools/probes/profile_pmc.sh C4 0 40 r4f "--dynamic 2" > $OUT/pmc_c4.log 2>&1; echo "pmc C4 done"
bash tools/probes/profile_pmc.sh C5 0 10 r4f "--dynamic 2" > $OUT/pmc_c5.log 2>&1; echo "pmc C5 done"
for wl in C2 C4 C5; do cp gpurun_out/pmc_r4f_$wl/summary.txt $OUT/pmc_${wl}_summary.txt; done
for a in "C5 8 40" "C5 2 40" "C4 2 100" "C2 2 400" "C5 8 40 one-stream"; do
  for sw in no_slab_overlap ""; do echo -n "[$sw] "; SPHX_DEBUG_SWITCHES=$sw timeout -k 10 200 python3 tools/probes/probe_slab_ring.py $a 2>&1 | grep -v amdgpu.ids; done
done | tee $OUT/slab_ring.txt
timeout -k 10 100 python3 tools/probes/probe_short_batch.py 2>&1 | grep -v amdgpu.ids | tee $OUT/short_batch.txt
