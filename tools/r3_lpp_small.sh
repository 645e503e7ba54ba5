#!/bin/bash
# small channels: compact kernels at 32 / 16 lanes per particle against the large-channel forms at 8 / 4
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_lppsmall; mkdir -p $O
run() { python bench.py --workload $1 --lpp $4 --steps $2 --warmup $3 --no-cpu-baseline --no-aux --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'lpp', d['config']['lanes_per_particle'], 'K', d['config']['rebuild_every'], f\"{1e3*d['ms_per_step']:.2f} us/step\")"; }
for rep in 1 2; do
for wl in C1 C2 "dp=0.02,DL=4" "dp=0.0125,DL=3"; do
  for l in 0 32 16 8 4; do run $wl 4000 400 $l; done
done; done 2>&1 | tee $O/lpp.txt
