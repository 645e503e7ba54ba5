#!/bin/bash
# counters behind the flat-load finding: FLAT / LDS instruction counts of the C5 passes, round-2 library against the current one
cd $GRAFT_REPO_ROOT
export PMC_GROUPS="SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"
SPHX_LIB=$GRAFT_REPO_ROOT/tools/_exp/libsphx_r2.so bash tools/probes/profile_pmc.sh C5 0 10 r3flat_old "--dynamic 2" > gpurun_out/r3_flat_old.log 2>&1
unset SPHX_LIB
bash tools/probes/profile_pmc.sh C5 0 10 r3flat_new "--dynamic 2" > gpurun_out/r3_flat_new.log 2>&1
for t in old new; do echo "== $t"; grep -E "^k_(continuity|kgc_w|forces_w|density_w)" gpurun_out/pmc_r3flat_${t}_C5/summary.txt | cut -c1-420; done
