#!/bin/bash
# ms/step of the large workloads over lanes-per-particle, with and without the LDS tile of the force pass (manual sweep)
for wl in C3 C4 C5; do
  case $wl in C3) st=1000;; C4) st=300;; C5) st=100;; esac
  for lpp in 1 2 4; do
    for lds in 1 0; do
      if [ $lds = 0 ]; then export SPHX_DEBUG_SWITCHES=no_lds_tiles; else unset SPHX_DEBUG_SWITCHES; fi
      python bench.py --workload $wl --lpp $lpp --steps $st --warmup 40 --no-aux --no-cpu-baseline --profile-steps 16 2>/dev/null | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl lpp=$lpp lds=$lds', round(1e3*d['ms_per_step'],1), 'us/step', d['kernels_ms'])"
    done
  done
done
