#!/bin/bash
# Round-3 evidence: bench lines, rocprofv3 kernel traces of the same commands, PMC passes (C2, C5), slab-ring probes.
cd $GRAFT_REPO_ROOT
TAG=${1:-r3}; OUT=gpurun_out/final_$TAG; mkdir -p $OUT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err; echo "driver-style rc=$?"
export TMPDIR=/tmp
for wl in C2 C4 C5; do
  case $wl in C2) st=4000; wu=400;; C4) st=300; wu=40;; C5) st=100; wu=40;; esac
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -- python3 bench.py --workload $wl \
      --steps $st --warmup $wu --no-cpu-baseline --no-aux --profile-steps 16 > $OUT/trace_$wl.json 2> $OUT/trace_$wl.err
  f=$(find $OUT/trace_$wl -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$wl.csv
  echo "trace $wl done"
done
bash tools/probes/profile_pmc.sh C2 0 200 ${TAG}f > $OUT/pmc_c2.log 2>&1; echo "pmc C2 done"
bash tools/probes/profile_pmc.sh C5 0 10 ${TAG}f "--dynamic 2" > $OUT/pmc_c5.log 2>&1; echo "pmc C5 done"
for a in "C5 8 40" "C5 2 40" "C4 2 100" "C2 2 400" "C5 8 40 one-stream" "C4 2 100 graph" "C2 2 400 graph"; do python tools/probes/probe_slab_ring.py $a; done 2>&1 | grep -v amdgpu.ids | tee $OUT/slab_ring.txt
python tools/probes/probe_short_batch.py 2>&1 | grep -v amdgpu.ids | tee $OUT/short_batch.txt
TAG=$TAG python - <<'PY'
import json, os
for f in ("bench", "bench_driver_style"):
    d = json.load(open(f"gpurun_out/final_{os.environ['TAG']}/{f}.json"))
    print(f, f"{d['value']:.4e}", f"{1e3*d['ms_per_step']:.2f} us/step", "roof", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4))
    for k, a in (d.get("aux") or {}).items():
        if "value" in a: print("  aux", k, f"{a['value']:.4e}", f"{1e3*a['ms_per_step']:.1f} us/step", "sustained", a.get("sustained") and f"{a['sustained']['value']:.4e}")
    if "accuracy" in d: print("  accuracy", d["accuracy"].get("L2"), d["accuracy"].get("steps"))
    if "cpu_baseline" in d: print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "x", d.get("gpu_over_cpu"))
PY
