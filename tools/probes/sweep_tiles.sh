#!/bin/bash
# manual sweep (not a test): LDS tile of the force pass on / off over channel sizes
cd "$(dirname "$0")/../.."
for spec in "dp=0.01,DL=9 1500" "dp=0.01,DL=12 1000" "dp=0.01,DL=18 800" "dp=0.01,DL=24 600" "dp=0.005,DL=9 400"; do
  set -- $spec
  for v in "" "SPHX_DEBUG_SWITCHES=no_lds_tiles"; do
    env $v timeout -k 10 200 python3 bench.py --workload $1 --steps $2 --warmup 80 --no-cpu-baseline --no-aux 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['config']
print('$1', '$v' or 'default', 'us/step %.2f'%(d['ms_per_step']*1e3), 'lpp', c['lanes_per_particle'], 'K', c['rebuild_every'], {k:round(v*1e3,1) for k,v in d['kernels_ms'].items() if k[:5] in ('k_kgc','k_for','k_con','k_den')}, flush=True)
"
  done
done
