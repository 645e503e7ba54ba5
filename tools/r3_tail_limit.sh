#!/bin/bash
# does the fused E|A launch + clock in the tail pay beyond 2 048 workgroups?  (mid-size channels: 262 k .. 1 M particles at 2 lanes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_tail; mkdir -p $O
run() { SPHX_DEBUG_SWITCHES=$1 python bench.py --workload $2 --steps $3 --warmup $4 --no-cpu-baseline --no-aux --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${1:-default}', '$2', f\"{1e3*d['ms_per_step']:.1f} us/step\")"; }
for rep in 1 2; do
for spec in "dp=0.01,DL=30 600 60" "C4 300 40" "dp=0.005,DL=20 300 40"; do
  set -- $spec
  run "" $1 $2 $3
  run tail_limit_8192 $1 $2 $3
done; done 2>&1 | tee $O/tail.txt
