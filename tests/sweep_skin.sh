#!/bin/bash
# manual tuning sweep (not a test): grid rebuild interval K x skin x workload
cd "$(dirname "$0")/.."
for spec in "C3 3 0 2000" "C3 5 0 2000" "C3 8 0 2000" "C4 1 0 300" "C4 3 0 300" "C4 5 0 300" "C4 8 0 300" "C5 1 0 40" "C5 3 0 40" "C5 5 0 40"; do
  set -- $spec
  timeout -k 10 200 python bench.py --workload $1 --rebuild-every $2 --skin $3 --steps $4 --warmup 40 --no-cpu-baseline --no-aux --profile-steps 20 2>gpurun_out/sweep_skin_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['config']
print('$1 K=$2 skin=$3h', 'value %.3e'%d['value'], 'us/step %.2f'%(d['ms_per_step']*1e3), c['cells'], c['rebuild_every'], 'skin %.4f'%c['skin'], 'forced', c['forced_rebuilds'], {k:round(v*1e3,1) for k,v in d['kernels_ms'].items()})
"
done
