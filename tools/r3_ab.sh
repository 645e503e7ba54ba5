#!/bin/bash
# A/B on one box: round-2 library (tools/_exp/libsphx_r2.so) against the current build, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_ab_${1:-x}; mkdir -p $O
run() { # name lib workload steps warmup
  SPHX_LIB=$2 python bench.py --workload $3 --steps $4 --warmup $5 --no-cpu-baseline --no-aux --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', '$3', f\"{1e3*d['ms_per_step']:.1f} us/step\")"
}
for rep in 1 2; do
for spec in "C2 4000 400" "C3 2000 200" "M194 1000 100" "C4 300 40" "C5 100 40"; do
  set -- $spec; wl=$1; [ $wl = M194 ] && wl="dp=0.01,DL=18"
  run old $GRAFT_REPO_ROOT/tools/_exp/libsphx_r2.so $wl $2 $3
  run new "" $wl $2 $3
done; done 2>&1 | tee $O/ab.txt
