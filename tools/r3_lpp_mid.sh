#!/bin/bash
# lanes per particle at 20 k .. 130 k particles with the round-3 kernels (fused E|A up to 4 096 workgroups)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_lppmid; mkdir -p $O
run() { python bench.py --workload $1 --lpp $4 --steps $2 --warmup $3 --no-cpu-baseline --no-aux --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'lpp', d['config']['lanes_per_particle'], 'K', d['config']['rebuild_every'], f\"{1e3*d['ms_per_step']:.1f} us/step\")"; }
for rep in 1 2; do
for wl in "dp=0.01,DL=2" "dp=0.01,DL=4" "C3" "dp=0.01,DL=9" "dp=0.01,DL=12"; do
  for l in 0 4 8 16; do run $wl 1500 150 $l; done
done; done 2>&1 | tee $O/lpp.txt
