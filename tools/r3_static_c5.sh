#!/bin/bash
# C5 on the static (host-scheduled) re-binning instead of the device-decided one, at several K / skins
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_static_c5; mkdir -p $O
for spec in "0 0 0.0" "2 8 0.0" "2 10 0.0" "2 8 0.6" "1 8 0.0"; do
  set -- $spec
  python - $1 $2 $3 <<'PY'
import importlib, sys
sys.path.insert(0, '.')
import bench
pkg = importlib.import_module(bench.PKG)
dyn, K, skin = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
r = bench.run_case(pkg.capi, pkg.config, pkg.geometry, "C5", dict(bench.WORKLOADS["C5"]), 100, 40, 0, rebuild_every=K, skin_h=skin, dynamic=dyn, sustained=(1000, 400))[0]
print("C5 dynamic", dyn, "K", r["tuning"]["rebuild_every"], "skin", round(r["tuning"]["skin"] / 0.0026, 3), "h", f"window {1e3*r['ms_per_step']:.1f} us/step  sustained {1e3*r['sustained']['ms_per_step']:.1f} us/step  forced/drift re-binnings {r['sustained']['forced_rebuilds']}", flush=True)
PY
done 2>&1 | grep -v amdgpu | tee $O/static.txt
