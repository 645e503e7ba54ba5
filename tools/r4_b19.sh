#!/bin/bash
# Round 4, batch 19: list words of pass A's walk addressed with scalar arithmetic (clamp by the array) on top of the near walks,
# against the build before both (tools/_exp/libsphx_r4k.so = commit cbfcec3); C5, C4, C3; then the suite
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4s; mkdir -p $OUT
timeout -k 10 600 python3 tools/probes/probe_ab_switches.py C5 100 40 1000 300 2 "@tools/_exp/libsphx_r4k.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/scalar_words_ab_c5.txt
timeout -k 10 300 python3 tools/probes/probe_ab_switches.py C4 300 40 2000 1000 2 "@tools/_exp/libsphx_r4k.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/scalar_words_ab_c4.txt
timeout -k 10 300 python3 tools/probes/probe_ab_switches.py C3 2000 100 0 0 2 "@tools/_exp/libsphx_r4k.so" "" 2>&1 | grep -v amdgpu.ids | tee $OUT/scalar_words_ab_c3.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.txt
