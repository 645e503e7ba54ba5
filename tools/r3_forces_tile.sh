#!/bin/bash
# tile size of the force pass with 88-byte records: 320 slots (five workgroups per CU) against the whole layout, 448 (four, no misses)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_forces_tile; mkdir -p $O
run() { SPHX_DEBUG_SWITCHES=$1 python bench.py --workload $2 --steps $3 --warmup $4 --no-cpu-baseline --no-aux --profile-steps 16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${1:-default}', '$2', f\"{1e3*d['ms_per_step']:.1f} us/step\", {k: round(v*1e3,1) for k,v in d['kernels_ms'].items() if k in ('k_kgc','k_continuity','k_density_walk','k_continuity_density','k_forces')})"; }
for rep in 1 2; do
for wl in "dp=0.01,DL=24" C4 "dp=0.005,DL=20"; do for sw in "" forces_tile_448; do run "$sw" $wl 300 40; done; done
for sw in forces_tile_320 ""; do run "$sw" C5 100 40; done
done > $O/times2.txt 2>&1
cat $O/times2.txt
