#!/bin/bash
# Round 4, first GPU pass: the whole -m gpu suite on the round's fixes, the two bench lines, and the C3 evidence
# (BASELINE.json configs[2]: "dp = 0.01 ... rocprof HBM GB/s") that round 3 left out -- kernel trace + PMC passes.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4a; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 420 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.txt
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-aux > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err; echo "driver-style rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_C3 -- python3 bench.py --workload C3 \
    --steps 2000 --warmup 100 --no-cpu-baseline --no-aux --profile-steps 64 > $OUT/trace_C3.json 2> $OUT/trace_C3.err
f=$(find $OUT/trace_C3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_C3.csv; rm -rf $OUT/trace_C3
echo "trace C3 done"
bash tools/probes/profile_pmc.sh C3 0 200 r4a > $OUT/pmc_c3.log 2>&1; echo "pmc C3 done"
cp gpurun_out/pmc_r4a_C3/summary.txt $OUT/pmc_c3_summary.txt
python3 - <<'PY'
import json
for f in ("bench", "bench_driver_style"):
    try:
        d = json.load(open(f"gpurun_out/r4a/{f}.json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, f"{d['value']:.4e}", f"{1e3*d['ms_per_step']:.2f} us/step", "roof", d["roofline"]["kernel"], round(d["roofline"]["frac"], 4))
    for k, a in (d.get("aux") or {}).items():
        if "value" in a: print("  aux", k, f"{a['value']:.4e}", f"{1e3*a['ms_per_step']:.1f} us/step", "roof", a["roofline"]["kernel"], round(a["roofline"]["frac"], 3), "sustained", a.get("sustained") and f"{a['sustained']['value']:.4e}")
        else: print("  aux", k, a)
    if "accuracy" in d: print("  accuracy", d["accuracy"])
    if "cpu_baseline" in d: print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "x", d.get("gpu_over_cpu"))
PY
