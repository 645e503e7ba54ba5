#!/bin/bash
# manual sweep (not a test): the default configuration over the workload sizes, with per-kernel eager timings
cd "$(dirname "$0")/../.."
for spec in "C2 2000" "C3 2000" "C4 300" "C5 40"; do
  set -- $spec
  timeout -k 10 200 python bench.py --workload $1 --steps $2 --warmup 40 --no-cpu-baseline --no-aux --profile-steps 20 2>gpurun_out/sweep_sizes_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['config']
print('$1', 'value %.3e'%d['value'], 'us/step %.2f'%(d['ms_per_step']*1e3), 'lpp', c['lanes_per_particle'], 'forced', c['forced_rebuilds'], {k:round(v*1e3,1) for k,v in d['kernels_ms'].items()})
"
done
