#!/bin/bash
# Round 4, batch 12: the decoupled re-binning policy -- GPU suite, bench lines, slab ring with the new default against K = 5
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4m; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.txt
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4m/bench.json"))
print("bench", f"{d['value']:.4e}", f"{1e3*d['ms_per_step']:.2f} us/step")
for k, a in (d.get("aux") or {}).items():
    if "value" in a: print("  aux", k, f"{a['value']:.4e}", f"{1e3*a['ms_per_step']:.1f} us/step", "K", a["tuning"]["rebuild_every"], "skin", a["tuning"]["skin"], "forced", a["tuning"]["forced_rebuilds"], "sustained", a.get("sustained") and (f"{a['sustained']['value']:.4e}", round(1e3*a['sustained']['ms_per_step'],1), a['sustained']['forced_rebuilds']))
    else: print("  aux", k, a)
print("  accuracy", {k: d["accuracy"].get(k) for k in ("L2", "L2_mean_profile_t16_20", "steps", "wall_seconds")})
PY
for a in "C5 8 40" "C5 8 40 K=5" "C4 2 100" "C4 2 100 K=5"; do SPHX_DEBUG_SWITCHES=no_slab_overlap timeout -k 10 200 python3 tools/probes/probe_slab_ring.py $a 2>&1 | grep -v amdgpu.ids; done | tee $OUT/slab_ring_k.txt
