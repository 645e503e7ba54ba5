#!/bin/bash
# PMC passes for the resident step (manual profiling helper, not a test).  Usage: profile_pmc.sh WORKLOAD LPP STEPS TAG
# (LPP 0 = auto).  Fold the result into profiles/ with tools/pmc_to_traffic.py.
# One rocprofv3 run per counter group (counters are never combined with sys/hip/hsa trace domains).
WL=${1:-C5}; LPP=${2:-0}; STEPS=${3:-20}; TAG=${4:-r1}; EXTRA=${5:-}   # EXTRA e.g. "--dynamic 2": static schedule, so that
# per-launch averages are not diluted by the launches that skip themselves in a dynamic context
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_${TAG}_${WL}"
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --workload $WL --lpp $LPP --steps $STEPS --warmup 4 --no-cpu-baseline --no-aux --profile-steps 0 $EXTRA"
i=0
# PMC_GROUPS="A B C;D E" overrides the counter groups (one rocprofv3 run per ';'-separated group)
if [ -n "$PMC_GROUPS" ]; then IFS=';' read -r -a GROUPS_ <<< "$PMC_GROUPS"; else GROUPS_=(); fi
for grp in "${GROUPS_[@]:-}" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  [ -z "$grp" ] && continue
  [ -n "$PMC_GROUPS" ] && [ $i -ge ${#GROUPS_[@]} ] && break
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- $CMD > "$OUT/g$i.json" 2> "$OUT/g$i.err" || echo "group $i failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(agg):
        line = k + " " + " ".join(f"{c}={v/max(calls[k][c],1):.4g}" for c, v in sorted(agg[k].items())) + f" launches={max(calls[k].values())}"
        print(line[:1500]); fo.write(line + "\n")
PY
