#!/bin/bash
# kernel stats of an in-process ring on ONE stream (per-kernel durations = one slab alone on the chip), old chain and second-stream form
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4_ring; mkdir -p $O
for sw in no_slab_overlap ""; do
  SPHX_DEBUG_SWITCHES=$sw timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${sw:-new} -- python3 tools/probes/probe_slab_ring.py C5 8 40 one-stream > $O/ring_${sw:-new}.txt 2>&1
  f=$(find $O/prof_${sw:-new} -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_${sw:-new}.csv; rm -rf $O/prof_${sw:-new}
  tail -1 $O/ring_${sw:-new}.txt
done
python3 - <<'PY'
import csv
for v in ("no_slab_overlap", "new"):
    print(v)
    for r in list(csv.DictReader(open(f"gpurun_out/r4_ring/kernel_stats_{v}.csv")))[:16]:
        print('   %-55s calls %6s avg %10.1f ns  tot %8.1f ms' % (r['Name'].split('(')[0][-55:], r['Calls'], float(r['AverageNs']), float(r['TotalDurationNs'])/1e6))
PY
