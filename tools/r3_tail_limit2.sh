#!/bin/bash
# with passes E|A fused up to 4 096 workgroups: 4 lanes per particle (fused up to 262 k particles) against 2 between 131 k and 262 k
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_tail; mkdir -p $O
run() { SPHX_DEBUG_SWITCHES=$1 python bench.py --workload $2 --lpp $5 --steps $3 --warmup $4 --no-cpu-baseline --no-aux --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${1:-default}', '$2', 'lpp', d['config']['lanes_per_particle'], f\"{1e3*d['ms_per_step']:.1f} us/step\")"; }
for rep in 1 2; do
for wl in "dp=0.01,DL=14" "dp=0.01,DL=18" "dp=0.01,DL=24"; do
  run tail_limit_4096 $wl 800 80 2
  run tail_limit_4096 $wl 800 80 4
done; done 2>&1 | tee $O/tail2.txt
