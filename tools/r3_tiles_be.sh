#!/bin/bash
# LDS tiles in passes B, E, A below 2 M particles, now that the tile is read with ds_read (it was read with flat loads)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_tiles_be; mkdir -p $O
run() { SPHX_DEBUG_SWITCHES=$1 python bench.py --workload $2 --steps $3 --warmup $4 --no-cpu-baseline --no-aux --profile-steps 16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${1:-default}', '$2', f\"{1e3*d['ms_per_step']:.1f} us/step\", {k: round(v*1e3,1) for k,v in d['kernels_ms'].items() if k in ('k_kgc','k_continuity','k_density_walk','k_continuity_density','k_forces')})"; }
for rep in 1 2; do
for spec in "C4 300 40" "dp=0.005,DL=20 300 40" "dp=0.004,DL=20 200 40" "dp=0.003,DL=24 150 40"; do
  set -- $spec
  run "" $1 $2 $3
  run tiles_be_from_1 $1 $2 $3
done; done 2>&1 | tee $O/tiles.txt
