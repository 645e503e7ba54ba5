#!/bin/bash
# Round-4 evidence, part 2 after the K / skin policy: PMC passes at C4 (its skin changed; C2, C3, C5 keep theirs -- the C5 passes
# with the skin and an interval that re-bins inside the short run), slab-ring probes, the short-batch probe
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4final; mkdir -p $OUT
bash tools/probes/profile_pmc.sh C4 0 40 r4g "--dynamic 2" > $OUT/pmc_c4.log 2>&1; echo "pmc C4 done"
bash tools/probes/profile_pmc.sh C5 0 10 r4g "--dynamic 2 --rebuild-every 6 --skin 0.28" > $OUT/pmc_c5.log 2>&1; echo "pmc C5 done"
for wl in C4 C5; do cp gpurun_out/pmc_r4g_$wl/summary.txt $OUT/pmc_${wl}_summary.txt; done
for a in "C5 8 40" "C5 2 40" "C4 2 100" "C2 2 400" "C5 8 40 one-stream"; do
  for ov in never always; do echo -n "[SPHX_SLAB_OVERLAP=$ov] "; SPHX_SLAB_OVERLAP=$ov timeout -k 10 200 python3 tools/probes/probe_slab_ring.py $a 2>&1 | grep -v amdgpu.ids; done
done | tee $OUT/slab_ring.txt
timeout -k 10 100 python3 tools/probes/probe_short_batch.py 2>&1 | grep -v amdgpu.ids | tee $OUT/short_batch.txt
