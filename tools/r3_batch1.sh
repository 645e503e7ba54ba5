#!/bin/bash
# (1) per-kernel A/B at C3 (round-2 library vs current)  (2) kernel trace of bench C5  (3) PMC passes at C5 (static schedule)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_batch1; mkdir -p $O
for lib in old new; do
  L=""; [ $lib = old ] && L=$GRAFT_REPO_ROOT/tools/_exp/libsphx_r2.so
  for wl in C3 "dp=0.01,DL=18"; do
  SPHX_LIB=$L python bench.py --workload $wl --steps 2000 --warmup 200 --no-cpu-baseline --no-aux --profile-steps 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', '$wl', f\"{1e3*d['ms_per_step']:.1f} us/step\", {k: round(v*1e3,1) for k,v in d['kernels_ms'].items()})"
  done
done 2>&1 | tee $O/c3_ab.txt
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_C5 -- python3 bench.py --workload C5 --steps 100 --warmup 40 --no-cpu-baseline --no-aux --profile-steps 16 > $O/trace_C5.json 2> $O/trace_C5.err
f=$(find $O/trace_C5 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_C5.csv && head -12 $O/kernel_stats_C5.csv | cut -c1-150
bash tools/probes/profile_pmc.sh C5 0 10 r3 "--dynamic 2" > $O/pmc_c5.log 2>&1; tail -12 $O/pmc_c5.log | cut -c1-400
