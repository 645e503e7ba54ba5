#!/bin/bash
# Round-4 evidence, part 1: the GPU suite, both bench lines, rocprofv3 kernel traces of the same commands (C2..C5)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4final; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 420 python3 -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.txt
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-aux > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err; echo "driver-style rc=$?"
for wl in C2 C3 C4 C5; do
  case $wl in C2) st=4000; wu=400;; C3) st=2000; wu=100;; C4) st=300; wu=40;; C5) st=100; wu=40;; esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -- python3 bench.py --workload $wl \
      --steps $st --warmup $wu --no-cpu-baseline --no-aux --profile-steps 16 > $OUT/trace_$wl.json 2> $OUT/trace_$wl.err
  f=$(find $OUT/trace_$wl -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$wl.csv; rm -rf $OUT/trace_$wl
  echo "trace $wl done"
done
python3 - <<'PY'
import json
for f in ("bench", "bench_driver_style"):
    d = json.load(open(f"gpurun_out/r4final/{f}.json"))
    print(f, f"{d['value']:.4e}", f"{1e3*d['ms_per_step']:.2f} us/step", "roof", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), "hbm_frac", d["roofline"].get("hbm_frac"))
    for k, a in (d.get("aux") or {}).items():
        if "value" in a: print("  aux", k, f"{a['value']:.4e}", f"{1e3*a['ms_per_step']:.1f} us/step", "roof", a["roofline"]["kernel"], round(a["roofline"]["frac"], 3), "hbm", a["roofline"].get("hbm_frac"), "sustained", a.get("sustained") and f"{a['sustained']['value']:.4e}")
        else: print("  aux", k, a)
    if "accuracy" in d: print("  accuracy", {k: d["accuracy"].get(k) for k in ("L2", "L2_mean_profile_t16_20", "steps", "wall_seconds")})
    if "cpu_baseline" in d: print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "x", d.get("gpu_over_cpu"))
PY
