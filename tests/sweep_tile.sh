#!/bin/bash
# manual tuning sweep (not a test): list-walking vs LDS-tiled kernels
cd "$(dirname "$0")/.."
for spec in "C3 4 -1 500" "C3 2 0 500" "C3 2 8 500" "C3 4 8 500" "C4 2 -1 100" "C4 2 0 100" "C4 1 0 100" "C4 4 0 100" "C4 2 8 100" "C5 1 -1 20" "C5 2 0 20" "C5 1 0 20" "C5 2 12 20" "C5 4 0 20"; do
  set -- $spec
  timeout -k 10 200 python bench.py --workload $1 --lpp $2 --tile $3 --steps $4 --warmup 20 --no-cpu-baseline --profile-steps 32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 lpp=$2 tile=$3', 'value %.3e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'step GB/s %.0f'%d['roofline']['step_achieved'], {k:round(v*1e3,1) for k,v in d['kernels_ms'].items()})
"
done
