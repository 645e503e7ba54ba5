#!/bin/bash
# force pass at a higher occupancy target (amdgpu_waves_per_eu) with a smaller LDS tile
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "" tools/_exp/libsphx_w6_t272.so tools/_exp/libsphx_w6_t256.so tools/_exp/libsphx_w7_t224.so; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
  SPHX_LIB=$L python bench.py --workload C5 --steps 100 --warmup 40 --no-cpu-baseline --no-aux --profile-steps 16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${lib:-default}', f\"{1e3*d['ms_per_step']:.1f} us/step\", 'forces', round(d['kernels_ms']['k_forces']*1e3,1))"
done; done
