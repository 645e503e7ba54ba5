#!/bin/bash
# Kernel-trace summary of one bench.py workload (manual profiling helper, not a test).  Usage: profile_trace.sh WORKLOAD STEPS TAG
# Writes gpurun_out/trace_<TAG>_<WL>/ (rocprofv3 --kernel-trace --stats, csv) and the bench line of the same command.
WL=${1:-C5}; STEPS=${2:-100}; TAG=${3:-r2}; WARM=${4:-40}
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="$ROOT/gpurun_out/trace_${TAG}_${WL}"
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --workload "$WL" \
    --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline --no-aux --profile-steps 16 > "$OUT/bench_line.json" 2> "$OUT/err.txt"
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -12 "$OUT/kernel_stats.csv"
