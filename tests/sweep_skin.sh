#!/bin/bash
# manual tuning sweep (not a test): grid rebuild interval K x lanes per particle x workload
cd "$(dirname "$0")/.."
for spec in "C2 5 32 2000" "C2 8 32 2000" "C2 10 32 2000" "C2 5 16 2000" "C3 5 4 2000" "C3 8 4 2000" "C3 5 8 2000" "C3 5 16 2000" "C4 5 4 300" "C4 8 4 300" "C4 5 2 300" "C4 5 8 300" "C5 5 4 40" "C5 8 4 40" "C5 5 2 40" "C5 5 8 40"; do
  set -- $spec
  timeout -k 10 200 python bench.py --workload $1 --rebuild-every $2 --lpp $3 --steps $4 --warmup 40 --no-cpu-baseline --no-aux --profile-steps 20 2>gpurun_out/sweep_skin_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['config']
print('$1 K=$2 lpp=$3', 'value %.3e'%d['value'], 'us/step %.2f'%(d['ms_per_step']*1e3), 'forced', c['forced_rebuilds'], {k:round(v*1e3,1) for k,v in d['kernels_ms'].items()})
"
done
