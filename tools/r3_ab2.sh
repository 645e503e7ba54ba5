#!/bin/bash
# A/B on one box: a saved build (tools/_exp/libsphx_base.so) against the current one, alternating, large channels only;
# with per-kernel times (bench.py --profile-steps) of the last repetition
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_ab2_${1:-x}; mkdir -p $O
run() { # name lib workload steps warmup
  SPHX_LIB=$2 python bench.py --workload $3 --steps $4 --warmup $5 --no-cpu-baseline --no-aux --profile-steps ${6:-0} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('kernels_ms') or {}
print('$1', '$3', f\"{1e3*d['ms_per_step']:.1f} us/step\", {a: round(1e3*b,1) for a,b in k.items()} if isinstance(k, dict) else '')"
}
for rep in 1 2 3; do
IFS=';' read -ra SPECLIST <<< "${SPECS:-C4 300 40;C5 100 40}"
for spec in "${SPECLIST[@]}"; do
  set -- $spec
  run old $GRAFT_REPO_ROOT/tools/_exp/libsphx_base.so $1 $2 $3 $([ $rep = 3 ] && echo 16)
  [ -n "$MID" ] && SPHX_DEBUG_SWITCHES=$MID run "new($MID)" "" $1 $2 $3 $([ $rep = 3 ] && echo 16)
  run new "" $1 $2 $3 $([ $rep = 3 ] && echo 16)
done; done > $O/ab.txt 2>&1
cat $O/ab.txt
