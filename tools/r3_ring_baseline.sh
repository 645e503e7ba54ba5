#!/bin/bash
# round 3: status quo of the in-process slab ring (tools/probes/probe_slab_ring.py) + kernel trace of the C5 G=8 ring
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_ring0
mkdir -p $O
python tools/probes/probe_slab_ring.py C5 8 40 > $O/ring.txt 2>&1
python tools/probes/probe_slab_ring.py C5 2 40 >> $O/ring.txt 2>&1
python tools/probes/probe_slab_ring.py C4 2 100 >> $O/ring.txt 2>&1
python tools/probes/probe_slab_ring.py C2 2 400 >> $O/ring.txt 2>&1
cat $O/ring.txt
rocprofv3 --kernel-trace --stats -d $O/prof -o c5g8 -- python3 tools/probes/probe_slab_ring.py C5 8 40 > $O/prof.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r3_ring0/prof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:40]:
        print(r['Name'][:70].ljust(70), r['Calls'].rjust(7), r['TotalDurationNs'].rjust(14), r['AverageNs'].rjust(12), r['Percentage'].rjust(7))
PY
