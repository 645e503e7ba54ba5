#!/bin/bash
# Round 4, batch 7: two launches per step on the small channels -- the GPU suite, then A/B against three launches and round 3
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout -k 10 420 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.txt
for wl in C2 C1 "dp=0.02,DL=4" "dp=0.015,DL=3"; do
timeout -k 10 200 python3 tools/probes/probe_ab_switches.py "$wl" 4000 400 0 0 3 "@tools/_exp/libsphx_r3.so" "no_two_launch" "" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_small.txt
done
timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-aux > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err; echo "driver-style rc=$?"
python3 -c "
import json; d=json.load(open('$OUT/bench_driver_style.json')); print('driver style', d['value'], d['ms_per_step']*1e3, d['roofline']['kernel'], d['roofline']['frac'], d['kernels_ms'])"
