#!/bin/bash
# Round 4, batch 10: the arming node inside exact-batch graphs -- GPU suite, then the driver's invocation against round 3 (alternating)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4k; mkdir -p $OUT
timeout -k 10 420 python3 -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.txt
for rep in 1 2 3 4; do
  for lib in tools/_exp/libsphx_r3.so ""; do
    SPHX_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 100 python3 bench.py --steps 20 --warmup 5 --no-aux --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[${lib:-new}]', f\"{d['value']:.4e}\", f\"{1e3*d['ms_per_step']:.2f} us/step\", d['config']['timed_slots'])"
  done
done | tee $OUT/driver_style_ab.txt
for rep in 1 2; do
  for lib in tools/_exp/libsphx_r3.so ""; do
    SPHX_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 100 python3 bench.py --steps 40 --warmup 5 --no-aux --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[${lib:-new}] 40 steps', f\"{d['value']:.4e}\", f\"{1e3*d['ms_per_step']:.2f} us/step\", d['config']['timed_slots'])"
  done
done | tee -a $OUT/driver_style_ab.txt
