#!/bin/bash
# manual tuning sweep (not a test): lanes-per-particle x workload
cd "$(dirname "$0")/../.."
for spec in "C2 8 2000" "C2 16 2000" "C2 32 2000" "C3 2 500" "C3 4 500" "C3 8 500" "C3 16 500" "C4 1 100" "C4 2 100" "C4 4 100" "C4 8 100" "C5 1 20" "C5 4 20"; do
  set -- $spec
  timeout -k 10 200 python bench.py --workload $1 --lpp $2 --steps $3 --warmup 20 --no-cpu-baseline --profile-steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 lpp=$2', 'value %.3e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'step GB/s %.0f'%d['roofline']['step_achieved'], {k:round(v*1e3,1) for k,v in d['kernels_ms'].items()})
"
done
