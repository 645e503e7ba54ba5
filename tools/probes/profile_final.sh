#!/bin/bash
# End-of-round evidence (manual): the bench lines, rocprofv3 kernel traces of the same commands, PMC passes of the headline workload.
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; TAG=${1:-r2}
OUT="$ROOT/gpurun_out/final_$TAG"; mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_style.json" 2> "$OUT/bench_driver_style.err"
cd /tmp; export TMPDIR=/tmp
for wl in C2 C4 C5; do
  case $wl in C2) st=4000; wu=400;; C4) st=300; wu=40;; C5) st=100; wu=40;; esac
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$wl" -- python3 "$ROOT/bench.py" --workload $wl \
      --steps $st --warmup $wu --no-cpu-baseline --no-aux --profile-steps 16 > "$OUT/trace_$wl.json" 2> "$OUT/trace_$wl.err"
  f=$(find "$OUT/trace_$wl" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$wl.csv"
done
"$ROOT/tools/probes/profile_pmc.sh" C2 0 200 ${TAG}f > "$OUT/pmc_c2.log" 2>&1
head -8 "$OUT/kernel_stats_C2.csv" | cut -c1-160
