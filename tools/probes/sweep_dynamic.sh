#!/bin/bash
# manual sweep (not a test): host-driven forced rebuilds (2) vs device-side re-bin decision (1), short and long windows
cd "$(dirname "$0")/../.."
for spec in "C3 2 6000" "C3 1 6000" "C4 2 300" "C4 1 300" "C4 2 6000" "C4 1 6000" "C5 2 40" "C5 1 40" "C5 2 1500" "C5 1 1500"; do
  set -- $spec
  timeout -k 10 300 python bench.py --workload $1 --dynamic $2 --steps $3 --warmup 40 --no-cpu-baseline --no-aux --profile-steps 0 2>gpurun_out/sweep_dyn_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('$1 dynamic=$2 steps=$3', 'value %.3e'%d['value'], 'us/step %.1f'%(d['ms_per_step']*1e3), 'rebuilds-by-drift/forced', c['forced_rebuilds'])
"
done
