#!/bin/bash
# Round 4, batch 4: the redirect / lean copy-back alone against round 3, full GPU suite, slab ring case
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4e; mkdir -p $OUT
timeout -k 10 420 python3 -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.txt
timeout -k 10 500 python3 tools/probes/probe_ab_switches.py C5 100 40 1000 300 3 "@tools/_exp/libsphx_r3.so" "@tools/_exp/libsphx_earlydt.so" "no_sched_redirect" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c5.txt
timeout -k 10 300 python3 tools/probes/probe_ab_switches.py C4 300 40 2000 1000 2 "@tools/_exp/libsphx_r3.so" "@tools/_exp/libsphx_earlydt.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c4.txt
timeout -k 10 100 python3 tools/probes/probe_ab_switches.py C3 2000 100 0 0 2 "@tools/_exp/libsphx_r3.so" "@tools/_exp/libsphx_earlydt.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c3.txt
timeout -k 10 100 python3 tools/probes/probe_ab_switches.py C2 4000 400 0 0 3 "@tools/_exp/libsphx_r3.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c2.txt
