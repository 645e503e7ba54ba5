#!/bin/bash
# us/step over the re-binning interval K for mid-size channels (manual sweep)
for wl in "dp=0.0125,DL=3" "dp=0.01,DL=3" "dp=0.01,DL=6" "dp=0.01,DL=12" "dp=0.005,DL=6"; do
  for k in 5 8 12 16; do
    python bench.py --workload $wl --rebuild-every $k --steps 1500 --warmup 80 --no-aux --no-cpu-baseline --profile-steps 0 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl K=$k', round(1e3*d['ms_per_step'],1), 'us/step  forced', d['config']['forced_rebuilds'], 'lpp', d['config']['lanes_per_particle'], d['config']['workload'].split('n_total=')[1].split(',')[0])"
  done
done
