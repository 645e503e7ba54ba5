#!/bin/bash
# A/B of an environment switch (manual experiment): sweep_ab.sh "VAR=1" [workloads]
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$ROOT"
SW=$1; shift
for wl in ${@:-C3 C4 C5}; do
  case $wl in C3) st=2000; wu=200;; C4) st=300; wu=40;; C5) st=100; wu=40;; *) st=1000; wu=100;; esac
  for v in "" "$SW"; do
    out=$(env $v python3 bench.py --workload $wl --steps $st --warmup $wu --no-cpu-baseline --no-aux 2>/dev/null | tail -1)
    python3 - "$wl" "$v" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[3])
k = d["kernels_ms"]
print(sys.argv[1], sys.argv[2] or "default", f"{d['ms_per_step']*1e3:.1f} us/step", f"{d['value']:.4g}",
      {n: round(v * 1e3, 1) for n, v in k.items() if n.startswith(("k_cont", "k_dens", "k_kgc", "k_forces"))}, flush=True)
PY
  done
done
