#!/bin/bash
# C2 / C1 headline: three launches per step (SPHX_DEBUG_SWITCHES=no_fuse_kgc) against two, alternating on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_c2ab_${1:-x}; mkdir -p $O
run() { # name switches workload steps warmup
  SPHX_DEBUG_SWITCHES=$2 python bench.py --workload $3 --steps $4 --warmup $5 --no-cpu-baseline --no-aux --profile-steps ${6:-0} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', '$3', '$4/$5', f\"{1e3*d['ms_per_step']:.2f} us/step\", {k: round(v*1e3,2) for k,v in d['kernels_ms'].items()})"
}
for rep in 1 2 3; do
for spec in "C2 4000 400" "C2 20 5" "C1 4000 400" "dp=0.02,DL=4 4000 400"; do
  set -- $spec
  run three no_fuse_kgc $1 $2 $3
  run two "" $1 $2 $3
done; done 2>&1 | tee $O/ab.txt
run three no_fuse_kgc C2 4000 400 200 | tee -a $O/ab.txt
run two "" C2 4000 400 200 | tee -a $O/ab.txt
