#!/bin/bash
# manual tuning sweep (not a test): grid rebuild interval K x workload (default lanes per particle)
cd "$(dirname "$0")/../.."
for spec in "C2 5 4000" "C2 8 4000" "C2 10 4000" "C2 16 4000" "C3 5 2000" "C3 8 2000" "C3 10 2000" "C4 5 300" "C4 8 300" "C4 10 300" "C5 5 40" "C5 8 40" "C5 10 40"; do
  set -- $spec
  timeout -k 10 200 python bench.py --workload $1 --rebuild-every $2 --steps $3 --warmup 80 --no-cpu-baseline --no-aux --profile-steps 20 2>gpurun_out/sweep_skin_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['config']
print('$1 K=$2', 'value %.3e'%d['value'], 'us/step %.2f'%(d['ms_per_step']*1e3), 'skin/h %.2f'%(c['skin']/(1.3*float(c['workload'].split('dp=')[1].split(',')[0]))), 'forced', c['forced_rebuilds'], {k:round(v*1e3,1) for k,v in d['kernels_ms'].items() if 'density' in k})
"
done
