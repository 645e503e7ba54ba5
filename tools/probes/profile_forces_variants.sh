#!/bin/bash
# Counter comparison of the force-pass variants (manual experiment, not a test): default | SPHX_FORCES=lds | SPHX_FORCES=pipe |
# SPHX_NO_PIPELINE=1 (round-1 kernel).  Usage: profile_forces_variants.sh WORKLOAD STEPS
WL=${1:-C4}; STEPS=${2:-20}
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_forces_${WL}"
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --workload $WL --steps $STEPS --warmup 4 --no-cpu-baseline --no-aux --profile-steps 0 --dynamic 2"
for variant in new lds pipe old; do
  case $variant in
    new) unset SPHX_FORCES SPHX_NO_PIPELINE;;
    lds) export SPHX_FORCES=lds; unset SPHX_NO_PIPELINE;;
    pipe) export SPHX_FORCES=pipe; unset SPHX_NO_PIPELINE;;
    old) unset SPHX_FORCES; export SPHX_NO_PIPELINE=1;;
  esac
  i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" \
             "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
             "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d "$OUT/$variant/g$i" -- $CMD > "$OUT/$variant.g$i.json" 2> "$OUT/$variant.g$i.err" || echo "$variant group $i failed"
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
with open(out + "/summary.txt", "w") as fo:
    for variant in ("new", "lds", "pipe", "old"):
        agg = collections.defaultdict(float); calls = collections.Counter()
        for f in glob.glob(f"{out}/{variant}/g*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_forces" not in r["Kernel_Name"]: continue
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]] += 1
        line = variant + " " + " ".join(f"{c}={v/max(calls[c],1):.4g}" for c, v in sorted(agg.items()))
        print(line); fo.write(line + "\n")
PY
