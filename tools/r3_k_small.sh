#!/bin/bash
# re-binning interval on the small channels with the round-3 lane counts
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_ksmall; mkdir -p $O
run() { python bench.py --workload $1 --rebuild-every $4 --steps $2 --warmup $3 --no-cpu-baseline --no-aux --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'lpp', d['config']['lanes_per_particle'], 'K', d['config']['rebuild_every'], 'forced', d['config']['forced_rebuilds'], f\"{1e3*d['ms_per_step']:.2f} us/step\")"; }
for rep in 1 2; do
for wl in C2 C1 "dp=0.02,DL=4"; do
  for K in 8 12 16 24 32 48; do run $wl 8000 800 $K; done
done; done 2>&1 | tee $O/k.txt
