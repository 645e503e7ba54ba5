#!/bin/bash
# large-channel bench lines (C3 / C4 / C5) with per-kernel times; TAG names the output directory
cd $GRAFT_REPO_ROOT
TAG=${1:-x}; O=gpurun_out/r3_bench_$TAG; mkdir -p $O
for spec in "C3 2000 200" "C4 300 40" "C5 100 40"; do
  set -- $spec
  python bench.py --workload $1 --steps $2 --warmup $3 --no-cpu-baseline --no-aux --profile-steps 16 > $O/$1.json 2> $O/$1.err || echo "$1 failed"
  python - $O/$1.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"]["workload"].split(":")[0], f"{1e3*d['ms_per_step']:.1f} us/step", f"{d['value']:.3e}", "roof", d["roofline"]["kernel"], f"{d['roofline']['launch_ms']*1e3:.1f} us frac {d['roofline']['frac']:.3f}",
      {k: round(v * 1e3, 1) for k, v in d["kernels_ms"].items() if v > 0.004})
PY
done
